% Builds the gateway against libndwt_hip.so (replaces mex/mex_compile.m:3 of the reference, which links FFTW).
% Run from the repository root on a machine with MATLAB and ROCm; the library itself is built by
%   make -C non-decimated_wavelets_amd/csrc        (hipcc --offload-arch=gfx950)
libdir = fullfile(pwd, 'non-decimated_wavelets_amd');
mex('-R2018a', '-v', fullfile('matlab', 'nd_dwt_hip_mex.c'), ['-I' fullfile(pwd, 'include')], ...
    ['-L' libdir], '-lndwt_hip', ['LDFLAGS=$LDFLAGS -Wl,-rpath,' libdir], '-outdir', 'matlab');
