% nd_dwt_4D_hip -- drop-in for the reference class Functions/nd_dwt_4D.m on the HIP engine (coefficients [n1, n2, n3, n4, 16 + 15 (level - 1)]).
%   nddwt = nd_dwt_4D_hip(wname, sizes, 'pres_l2_norm', 1, 'precision', 'single');
%   y = nddwt.dec(x, level);   x_hat = nddwt.rec(y);
% Everything shared lives in nd_dwt_hip_base.m; this file holds what differs between the four reference classes.
classdef nd_dwt_4D_hip < nd_dwt_hip_base
    methods
        function obj = nd_dwt_4D_hip(wname, sizes, varargin)
            obj = obj.init_(wname, sizes, varargin{:});
        end
    end
    methods (Access = protected)
        function d = ndim_(~), d = 4; end
        function msg = size_error_(~), msg = 'The sizes vector must be length 4'; end      % nd_dwt_4D.m:82
        function msg = wname_error_(~), msg = 'You must specify four filter names in a cell array of length 4, or a single string for the same filter to be used in all dimensions'; end    % nd_dwt_4D.m:93-95
        function level = level_from_bands_(~, nb), level = 1 + (nb - 16) / 15; end      % nd_dwt_4D.m:213
    end
end
