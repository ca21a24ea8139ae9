% nd_dwt_2D_hip -- drop-in for the reference class Functions/nd_dwt_2D.m on the HIP engine (coefficients [n1, n2, 4 + 3 (level - 1)]).
%   nddwt = nd_dwt_2D_hip(wname, sizes, 'pres_l2_norm', 1, 'precision', 'single');
%   y = nddwt.dec(x, level);   x_hat = nddwt.rec(y);
% Everything shared lives in nd_dwt_hip_base.m; this file holds what differs between the four reference classes.
classdef nd_dwt_2D_hip < nd_dwt_hip_base
    methods
        function obj = nd_dwt_2D_hip(wname, sizes, varargin)
            obj = obj.init_(wname, sizes, varargin{:});
        end
    end
    methods (Access = protected)
        function d = ndim_(~), d = 2; end
        function msg = size_error_(~), msg = 'The sizes vector must be length 2'; end      % nd_dwt_2D.m:81
        function msg = wname_error_(~), msg = 'You must specify two filter names in a cell array of length 2, or a single string for the same filter to be used in all dimensions'; end    % nd_dwt_2D.m:92-94
        function level = level_from_bands_(~, nb), level = 1 + (nb - 4) / 3; end      % nd_dwt_2D.m:215
    end
end
