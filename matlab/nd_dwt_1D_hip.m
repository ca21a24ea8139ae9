% nd_dwt_1D_hip -- drop-in for the reference class Functions/nd_dwt_1D.m on the HIP engine (coefficients [n, 1 + level]).
%   nddwt = nd_dwt_1D_hip(wname, sizes, 'pres_l2_norm', 1, 'precision', 'single');
%   y = nddwt.dec(x, level);   x_hat = nddwt.rec(y);
% Everything shared lives in nd_dwt_hip_base.m; this file holds what differs between the four reference classes.
classdef nd_dwt_1D_hip < nd_dwt_hip_base
    methods
        function obj = nd_dwt_1D_hip(wname, sizes, varargin)
            obj = obj.init_(wname, sizes, varargin{:});
        end
    end
    methods (Access = protected)
        function d = ndim_(~), d = 1; end
        function msg = size_error_(~), msg = '1D array length must be a scalar'; end      % nd_dwt_1D.m:88
        function msg = wname_error_(~), msg = 'Wavelet Name Must be a string'; end    % nd_dwt_1D.m:84
        function level = level_from_bands_(~, nb), level = ceil(nb - 1); end      % nd_dwt_1D.m:213
    end
end
