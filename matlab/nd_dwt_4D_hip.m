% nd_dwt_4D_hip -- the reference class nd_dwt_4D with compute = 'hip': same constructor, dec(x,level), rec(y) and
% properties (Functions/nd_dwt_4D.m); the transform runs on the MI355X through nd_dwt_hip_mex -> libndwt_hip.so.
% No FFT-domain kernels are built (f_dec stays empty): the engine needs only the wavelet names.
%
%   nddwt = nd_dwt_4D_hip(wname, sizes, 'pres_l2_norm', 1, 'precision', 'single');
%   y = nddwt.dec(x, level);   x_hat = nddwt.rec(y);
classdef nd_dwt_4D_hip
    properties
        f_dec = [];     % not materialised by this backend
        sizes; f_size; wname; pres_l2_norm = 0; compute = 'hip'; precision = 'double'; dilation = 'reference';
    end
    methods
        function obj = nd_dwt_4D_hip(wname, sizes, varargin)
            if length(sizes) ~= 4, error('The sizes vector must be length 4'); end
            obj.sizes = sizes;
            if ischar(wname), obj.wname = repmat({wname}, 1, max(4, 2)); else, obj.wname = wname; end
            if mod(length(varargin), 2), error('Optional inputs must come in pairs'); end
            for ind = 1:2:length(varargin)          % same keys as nd_dwt_3D.m:109-120
                switch lower(varargin{ind})
                    case 'pres_l2_norm', obj.pres_l2_norm = varargin{ind+1};
                    case 'compute',      obj.compute = varargin{ind+1};
                    case 'precision',    obj.precision = varargin{ind+1};
                    case 'dilation',     obj.dilation = varargin{ind+1};
                    otherwise, warning(sprintf('Unknown optional input #%d ingoring!', ind));
                end
            end
        end
        function y = dec(obj, x, level)
            if 4 == 1 && size(x, 1) == 1, x = x.'; end               % nd_dwt_1D.m:151-153
            if strcmpi(obj.precision, 'single'), x = single(x); else, x = double(x); end
            y = nd_dwt_hip_mex(x, obj.wname, 0, level, obj.pres_l2_norm, obj.dilation);
        end
        function y = rec(obj, x)
            nb = size(x, 4 + 1);
            level = 1 + (nb - 2^4) / (2^4 - 1);                    % nd_dwt_2D.m:215 / nd_dwt_4D.m:213 form
            if strcmpi(obj.precision, 'single'), x = single(x); else, x = double(x); end
            y = nd_dwt_hip_mex(x, obj.wname, 1, level, obj.pres_l2_norm, obj.dilation);
        end
    end
end
