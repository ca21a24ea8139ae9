% example_nd_dwt_hip.m -- the nd_dwt_{1..4}D_hip classes in the same walk-through as the reference's
% example_nd_dwt_{1,2,3,4}D.m (random complex signal, dec, rec, energies, reconstruction error).
% Not run in the build container (no MATLAB there).  Build the gateway first: run matlab/ndwt_hip_compile.m
addpath(fileparts(mfilename('fullpath')));
cases = { {@nd_dwt_1D_hip, 1024,            'db3'},                     ...
          {@nd_dwt_2D_hip, [256 256],       {'db1','db4'}},             ...
          {@nd_dwt_3D_hip, [64 64 20],      {'db1','db3','db9'}},       ...
          {@nd_dwt_4D_hip, [32 32 16 16],   {'db1','db3','db2','db4'}} };
level = 2;
for k = 1:numel(cases)
    c = cases{k};
    sz = c{2};
    if isscalar(sz), x = randn(sz,1) + 1j*randn(sz,1); else, x = randn(sz) + 1j*randn(sz); end
    nddwt   = c{1}(c{3}, sz, 'pres_l2_norm', true);
    x_trans = nddwt.dec(x, level);
    x_recon = nddwt.rec(x_trans);
    fprintf('%d-D: Energy in signal domain = %g \t Energy in wavelet domain = %g\n', k, norm(x(:)), norm(x_trans(:)));
    fprintf('     Absolute max reconstruction error = %g\n', max(abs(x_recon(:) - x(:))));
end
