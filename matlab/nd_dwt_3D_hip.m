% nd_dwt_3D_hip -- drop-in for the reference class Functions/nd_dwt_3D.m on the HIP engine (coefficients [n1, n2, n3, 8 + 7 (level - 1)]).
%   nddwt = nd_dwt_3D_hip(wname, sizes, 'pres_l2_norm', 1, 'precision', 'single');
%   y = nddwt.dec(x, level);   x_hat = nddwt.rec(y);
% Everything shared lives in nd_dwt_hip_base.m; this file holds what differs between the four reference classes.
classdef nd_dwt_3D_hip < nd_dwt_hip_base
    methods
        function obj = nd_dwt_3D_hip(wname, sizes, varargin)
            obj = obj.init_(wname, sizes, varargin{:});
        end
    end
    methods (Access = protected)
        function d = ndim_(~), d = 3; end
        function msg = size_error_(~), msg = 'The sizes vector must be length 3'; end      % nd_dwt_3D.m:83
        function msg = wname_error_(~), msg = 'You must specify three filter names in a cell arrayof length 3, or a single string for the same filter to be used in all dimensions'; end    % nd_dwt_3D.m:94-96
        function level = level_from_bands_(~, nb), level = ceil(nb / 8); end      % nd_dwt_3D.m:217
    end
end
