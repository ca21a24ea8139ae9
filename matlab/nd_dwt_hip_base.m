% nd_dwt_hip_base -- shared implementation of the four drop-in classes nd_dwt_{1,2,3,4}D_hip: the reference classes
% Functions/nd_dwt_{1,2,3,4}D.m with compute = 'hip'.  Same constructor (name/value pairs 'pres_l2_norm', 'compute',
% 'precision'; unknown keys warn), same public properties (f_dec, sizes, f_size, wname, pres_l2_norm, compute, precision),
% dec(x, level), rec(y) with the per-class level inference; the transform runs on the MI355X through
% nd_dwt_hip_mex -> libndwt_hip.so.  No FFT-domain kernels are built: f_dec holds the per-axis taps {LO_D, HI_D} instead of
% 2^d full-size volumes (the engine only needs the wavelet names).  The subclasses supply the dimension count, the
% reference's error messages and its level formula.
classdef (Abstract) nd_dwt_hip_base
    properties
        f_dec           % per-axis taps {LO_D, HI_D} (wave_filters); the reference's N-D DFT kernels are never materialised
        sizes           % size of the arrays the object transforms
        f_size          % filter lengths, fields s1 .. sd (nd_dwt_3D.m:271-275)
        wname           % cell array of wavelet names, one per dimension
        pres_l2_norm = 0;
        compute = 'hip';
        precision = 'double';
        dilation = 'reference';     % 'atrous': textbook SWT (taps dilated by 2^(level-1)); the reference never dilates
        devices = [];               % HIP device ordinals: empty = device 0, one = that device, several = the outermost axis sharded over them (2-D .. 4-D)
        exchange = 'scatter';       % several devices, rec: 'scatter' = one band of partial sums per level (equal to one device to rounding); 'gather' = bit-identical
    end
    methods (Abstract, Access = protected)
        d = ndim_(obj)                       % number of dimensions
        msg = size_error_(obj)               % message of the sizes check (nd_dwt_3D.m:83)
        msg = wname_error_(obj)              % message of the wavelet-name count check (nd_dwt_3D.m:94-96)
        level = level_from_bands_(obj, nb)   % level inference of rec() (nd_dwt_1D.m:213, 2D:215, 3D:217, 4D:213)
    end
    methods
        function obj = init_(obj, wname, sizes, varargin)
            d = obj.ndim_();
            if length(sizes) ~= d, error(obj.size_error_()); end
            obj.sizes = sizes;
            if ischar(wname)
                obj.wname = repmat({wname}, 1, max(d, 2));           % nd_dwt_1D.m keeps two names as well
            elseif d == 1 || length(wname) ~= d                       % 1-D: the name must be a string (nd_dwt_1D.m:84)
                error(obj.wname_error_());
            else
                obj.wname = wname;
            end
            if mod(length(varargin), 2), error('Optional inputs must come in pairs'); end
            for ind = 1:2:length(varargin)                            % nd_dwt_3D.m:109-120
                switch lower(varargin{ind})
                    case 'pres_l2_norm', obj.pres_l2_norm = varargin{ind+1};
                    case 'compute',      obj.compute = varargin{ind+1};
                    case 'precision',    obj.precision = varargin{ind+1};
                    case 'dilation',     obj.dilation = varargin{ind+1};
                    case 'devices',      obj.devices = double(varargin{ind+1}(:).');
                    case 'exchange',     obj.exchange = varargin{ind+1};
                    otherwise, warning(sprintf('Unknown optional input #%d ingoring!', ind));
                end
            end
            % get_filters (nd_dwt_3D.m:263-286): taps per axis and the length check against the data size
            ord = {'First', 'Second', 'Third', 'Fourth'};
            obj.f_dec = cell(1, d);
            for a = 1:d
                [lo_d, hi_d] = wave_filters(obj.wname{a});
                obj.f_dec{a} = {lo_d, hi_d};
                obj.f_size.(sprintf('s%d', a)) = length(lo_d);
                if length(lo_d) > sizes(a)
                    error([ord{a} ' Dimension of Data is shorter than the wavelet filter being used']);
                end
            end
        end
        function y = dec(obj, x, level)
            if obj.ndim_() == 1 && size(x, 1) == 1, x = x.'; end      % nd_dwt_1D.m:151-153
            if strcmpi(obj.precision, 'single'), x = single(x); else, x = double(x); end
            y = nd_dwt_hip_mex(x, obj.wname(1:obj.ndim_()), 0, level, obj.pres_l2_norm, obj.dilation, obj.devices, obj.exchange);
        end
        % ---- device-resident coefficients: only the signal crosses PCIe (the reference's dec / rec move the whole coefficient array) ----
        function h = dec_keep(obj, x, level)
            % dec(x, level) whose result stays on the device; returns a uint64 handle for rec_kept / shrink_kept / fetch / release
            if obj.ndim_() == 1 && size(x, 1) == 1, x = x.'; end
            if strcmpi(obj.precision, 'single'), x = single(x); else, x = double(x); end
            h = nd_dwt_hip_mex('dec_keep', x, obj.wname(1:obj.ndim_()), level, obj.pres_l2_norm, obj.dilation, obj.device_());
        end
        function x = rec_kept(obj, h),            x = nd_dwt_hip_mex('rec_handle', h); end   %#ok<INUSL>
        function shrink_kept(obj, h, thr, mode)   %#ok<INUSL>
            if nargin < 4, mode = 'soft'; end
            nd_dwt_hip_mex('shrink', h, thr, mode);
        end
        function y = fetch(obj, h),               y = nd_dwt_hip_mex('fetch', h); end        %#ok<INUSL>
        function release(obj, h),                 nd_dwt_hip_mex('release', h); end          %#ok<INUSL>
        function xd = denoise(obj, x, level, thr, mode)
            % rec(shrink(dec(x, level), thr)) in one call; the coefficients never leave the device
            if nargin < 5, mode = 'soft'; end
            if obj.ndim_() == 1 && size(x, 1) == 1, x = x.'; end
            if strcmpi(obj.precision, 'single'), x = single(x); else, x = double(x); end
            xd = nd_dwt_hip_mex('denoise', x, obj.wname(1:obj.ndim_()), level, obj.pres_l2_norm, thr, mode, obj.dilation, obj.device_());
        end
        function dev = device_(obj)
            if isempty(obj.devices), dev = 0; else, dev = obj.devices(1); end
        end
        function y = rec(obj, x)
            level = obj.level_from_bands_(size(x, obj.ndim_() + 1));
            if strcmpi(obj.precision, 'single'), x = single(x); else, x = double(x); end
            y = nd_dwt_hip_mex(x, obj.wname(1:obj.ndim_()), 1, level, obj.pres_l2_norm, obj.dilation, obj.devices, obj.exchange);
        end
    end
end
