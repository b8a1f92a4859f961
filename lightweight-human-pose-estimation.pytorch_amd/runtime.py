"""Host-side engine over one C-ABI handle: marshals numpy / torch buffers to the library.

PyTorch is used only as a tensor container (device memory, ``data_ptr()``); every computation
is a HIP kernel behind include/lwpose.h.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, check, lib

_ROLE_NBT = 6


def _torch():
    import torch
    return torch


class Engine(object):
    def __init__(self, device_id=0, nref=1, num_channels=128, num_heatmaps=19, num_pafs=38, dtype=_lib.F32):
        self.h = _lib.Handle(device_id, nref, num_channels, num_heatmaps, num_pafs, dtype)
        self.nref, self.C, self.NH, self.NP = nref, num_channels, num_heatmaps, num_pafs
        self.device_id = device_id
        self._keep = None

    # ------------------------------------------------------------------ stream ordering
    def _order(self, device=None, hand_over=True):
        """Order the library's stream against torch's CURRENT stream on this GPU with events (lwp_set_stream): no host block.
        The reference's net(x) runs on the current stream (demo.py:64-68); this gives device tensors the same semantics —
        inputs written by queued torch work are waited for, and torch work queued after the call sees the device results.
        ``hand_over=False``: results left on the device stay on the library's stream (for its own next call)."""
        torch = _torch()
        st = torch.cuda.current_stream(torch.device("cuda", self.device_id) if device is None else device)
        check(lib().lwp_set_stream(self.h.ptr, C.c_void_p(st.cuda_stream), 1 if hand_over else 2), self.h.ptr)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict):
        """state_dict: key -> torch tensor / numpy array (float32; num_batches_tracked int64)."""
        names, arrs = [], []
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            if a.dtype != np.int64:
                a = np.ascontiguousarray(a, dtype=np.float32)
            names.append(k.encode())
            arrs.append(np.ascontiguousarray(a))
        n = len(names)
        c_names = (C.c_char_p * n)(*names)
        c_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        shapes = np.ones((n, 4), dtype=np.int64)
        ndims = np.zeros(n, dtype=np.int32)
        for i, a in enumerate(arrs):
            ndims[i] = a.ndim
            shapes[i, :a.ndim] = a.shape
        check(lib().lwp_load_weights(self.h.ptr, c_names, c_ptrs, shapes.ctypes.data_as(C.POINTER(C.c_int64)),
                                     ndims.ctypes.data_as(C.POINTER(C.c_int)), n), self.h.ptr)

    def weights_blob_bytes(self):
        n = C.c_size_t()
        check(lib().lwp_weights_blob_bytes(self.h.ptr, C.byref(n)), self.h.ptr)
        return n.value

    def export_weights(self, device_tensor):
        check(lib().lwp_weights_blob_export(self.h.ptr, device_tensor.data_ptr(), device_tensor.numel() * device_tensor.element_size()), self.h.ptr)

    def import_weights(self, device_tensor):
        check(lib().lwp_weights_blob_import(self.h.ptr, device_tensor.data_ptr(), device_tensor.numel() * device_tensor.element_size()), self.h.ptr)

    def set_capacity(self, max_peaks=2048, max_kpts=128, max_conn=4096, max_entries=256):
        check(lib().lwp_set_capacity(self.h.ptr, max_peaks, max_kpts, max_conn, max_entries), self.h.ptr)
        self._caps = (max_peaks, max_kpts, max_conn, max_entries)

    @property
    def caps(self):
        return getattr(self, "_caps", (2048, 128, 4096, 256))

    # ------------------------------------------------------------------ network
    def forward(self, x):
        """x: (N,3,H,W) float32 torch tensor (cpu or cuda) or numpy array -> list of 2(1+nref) outputs of the
        same kind (NCHW), like PoseEstimationWithMobileNet.forward (with_mobilenet.py:114-123)."""
        torch = _torch()
        is_np = isinstance(x, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if is_np else x
        if t.dim() != 4 or t.shape[1] != 3:
            raise ValueError("expected input of shape (N, 3, H, W), got %s" % (tuple(t.shape),))
        t = t.detach().to(torch.float32).contiguous()
        N, _, H, W = t.shape
        on_dev = t.is_cuda
        if on_dev and t.device.index != self.device_id:
            raise ValueError("input is on cuda:%d but the engine lives on cuda:%d" % (t.device.index, self.device_id))
        fh, fw = H, W
        for _ in range(3):                       # three stride-2 stages: out = (in - 1) // 2 + 1
            fh, fw = (fh - 1) // 2 + 1, (fw - 1) // 2 + 1
        shapes = [(N, self.NP if i % 2 else self.NH, fh, fw) for i in range(2 * (1 + self.nref))]
        outs = [torch.empty(s, dtype=torch.float32, device=t.device) for s in shapes]
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        if on_dev:
            self._order(t.device)                 # events both ways: the outputs are valid for work queued on torch's current stream
            # (a frame from preprocess_u8(hand_over=False) is in order on the engine's stream; the outputs still need the hand-over)
        mem = MEM_DEVICE if on_dev else MEM_HOST
        check(lib().lwp_forward(self.h.ptr, t.data_ptr(), mem, N, H, W, ptrs, mem), self.h.ptr)
        return [o.numpy() for o in outs] if is_np else outs

    def synchronize(self):
        check(lib().lwp_synchronize(self.h.ptr), self.h.ptr)

    # ------------------------------------------------------------------ post-processing pieces
    def upsample(self, maps_nchw, ratio=4):
        """(N,C,h,w) float32 numpy -> (N, h*r, w*r, C) float32 numpy (cv2.resize INTER_CUBIC, demo.py:72,76)."""
        if getattr(maps_nchw, "is_cuda", False):
            a = maps_nchw.detach().contiguous()
            self._order(a.device)
            ptr, mem = a.data_ptr(), MEM_DEVICE
        else:
            a = np.ascontiguousarray(maps_nchw.numpy() if hasattr(maps_nchw, "numpy") else maps_nchw, dtype=np.float32)
            ptr, mem = a.ctypes.data, MEM_HOST
        N, Cc, h, w = a.shape
        out = np.empty((N, h * ratio, w * ratio, Cc), dtype=np.float32)
        check(lib().lwp_upsample(self.h.ptr, ptr, mem, N, Cc, h, w, ratio, out.ctypes.data, MEM_HOST), self.h.ptr)
        return out

    @staticmethod
    def preprocess_dims(height, width, net_input_height_size, stride):
        """(scaled_h, scaled_w, out_h, out_w, pad [top,left,bottom,right], scale) of demo.py:55-62 for a height x width frame."""
        v = [C.c_int() for _ in range(4)]
        pad = (C.c_int * 4)()
        sc = C.c_double()
        check(lib().lwp_preprocess_dims(height, width, net_input_height_size, stride, *[C.byref(a) for a in v], pad, C.byref(sc)))
        return v[0].value, v[1].value, v[2].value, v[3].value, [int(a) for a in pad], sc.value

    def preprocess_u8(self, img, net_input_height_size, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1 / 256,
                      hand_over=True):
        """uint8 HxWx3 frame (numpy or cuda tensor) -> (x: 1x3xH'xW' float32 cuda tensor, scale, pad): the cubic resize,
        normalize and pad_width of demo.py:55-64 in one kernel.  ``hand_over=False`` (infer_fast's internal use): x is only meant
        for this engine's next call — it stays on the engine's stream and that call skips the stream ordering (no events)."""
        torch = _torch()
        if getattr(img, "is_cuda", False):
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
                raise TypeError("frame must be HxWx3 uint8")
            a = img.contiguous()
            ptr, mem = a.data_ptr(), MEM_DEVICE
        else:
            a = np.ascontiguousarray(img)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise TypeError("frame must be HxWx3 uint8")
            ptr, mem = a.ctypes.data, MEM_HOST
        H, W = int(a.shape[0]), int(a.shape[1])
        _, _, oh, ow, pad, scale = self.preprocess_dims(H, W, net_input_height_size, stride)
        x = torch.empty((1, 3, oh, ow), dtype=torch.float32, device=torch.device("cuda", self.device_id))
        pv = (C.c_double * 3)(*[float(v) for v in pad_value])
        mv = (C.c_double * 3)(*[float(v) for v in img_mean])
        self._order(hand_over=hand_over)          # x is handed to torch's current stream by event; a host frame may be reused on return
        check(lib().lwp_preprocess_u8(self.h.ptr, ptr, mem, H, W, net_input_height_size, stride, pv, mv, float(img_scale), x.data_ptr()), self.h.ptr)
        if not hand_over:
            x._lwp_stream_owner = self            # produced on this engine's stream, not visible to torch's stream
        return x, scale, pad

    @staticmethod
    def scale_dims(height, width, ratio, base_height, stride):
        """(scaled_h, scaled_w, out_h, out_w, pad [top,left,bottom,right]) of val.py:89-91 for a height x width frame."""
        v = [C.c_int() for _ in range(4)]
        pad = (C.c_int * 4)()
        check(lib().lwp_scale_dims(height, width, float(ratio), base_height, stride, *[C.byref(a) for a in v], pad))
        return v[0].value, v[1].value, v[2].value, v[3].value, [int(a) for a in pad]

    def preprocess_scaled_u8(self, imgs, ratio, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1 / 256):
        """N same-sized frames (N,H,W,3) or one (H,W,3), numpy or cuda tensor, uint8 or float32 -> (x: N x 3 x H' x W' float32
        cuda tensor, pad): normalize + cubic resize by ``ratio`` + pad_width of val.py:84-93 in one kernel."""
        torch = _torch()
        on_dev = getattr(imgs, "is_cuda", False)
        a = imgs.contiguous() if on_dev else np.ascontiguousarray(imgs)
        is_u8 = a.dtype == (torch.uint8 if on_dev else np.uint8)
        is_f32 = a.dtype == (torch.float32 if on_dev else np.float32)
        if not (is_u8 or is_f32) or len(a.shape) not in (3, 4) or a.shape[-1] != 3:
            raise TypeError("frames must be (N,)HxWx3 uint8 or float32")
        shp = tuple(a.shape) if len(a.shape) == 4 else (1,) + tuple(a.shape)
        N, H, W = int(shp[0]), int(shp[1]), int(shp[2])
        _, _, oh, ow, pad = self.scale_dims(H, W, ratio, base_height, stride)
        if on_dev:
            if a.device.index != self.device_id:
                raise ValueError("frames are on cuda:%d but the engine lives on cuda:%d" % (a.device.index, self.device_id))
            ptr, mem = a.data_ptr(), MEM_DEVICE
        else:
            ptr, mem = a.ctypes.data, MEM_HOST
        self._order()
        x = torch.empty((N, 3, oh, ow), dtype=torch.float32, device=torch.device("cuda", self.device_id))
        pv = (C.c_double * 3)(*[float(v) for v in pad_value])
        mv = (C.c_double * 3)(*[float(v) for v in img_mean])
        fn = lib().lwp_preprocess_scaled_u8 if is_u8 else lib().lwp_preprocess_scaled_f32
        check(fn(self.h.ptr, ptr, mem, N, H, W, float(ratio), base_height, stride, pv, mv, float(img_scale), x.data_ptr()), self.h.ptr)
        return x, pad

    def multiscale_accumulate(self, accum, maps, up_ratio, pad, n_scales, init=False):
        """accum (H,W,C) or (N,H,W,C) float32 [numpy or cuda tensor, updated in place] += resize(crop(upsample(maps))) / n_scales
        (val.py:96-101).  maps: (C,h,w) or (N,C,h,w) float32 numpy / cuda tensor; the N frames share pad and size.
        init=True: accum is treated as zero (first scale), so it may be uninitialised memory."""
        def ptr_mem(a):
            if getattr(a, "is_cuda", False):
                return a.data_ptr(), MEM_DEVICE
            return a.ctypes.data, MEM_HOST
        if getattr(maps, "is_cuda", False):
            maps = maps.detach().contiguous()
        else:
            maps = np.ascontiguousarray(maps, dtype=np.float32)
        shp = tuple(maps.shape)
        ashp = tuple(accum.shape)
        N = shp[0] if len(shp) == 4 else 1
        if len(ashp) not in (3, 4) or (len(ashp) == 4 and ashp[0] != N) or (len(ashp) == 3 and N != 1):
            raise ValueError("accum / maps batch mismatch")
        H, W, Cc = ashp[-3:]
        if Cc != shp[-3]:
            raise ValueError("channel mismatch")
        if getattr(accum, "is_cuda", False):
            if not accum.is_contiguous() or accum.dtype != _torch().float32:
                raise TypeError("accum must be a contiguous float32 tensor")
        elif accum.dtype != np.float32 or not accum.flags.c_contiguous:
            raise TypeError("accum must be a C-contiguous float32 array")
        mp, mm = ptr_mem(maps)
        ap, am = ptr_mem(accum)
        if mm == MEM_DEVICE or am == MEM_DEVICE:
            self._order()
        padv = (C.c_int * 4)(*[int(v) for v in pad])
        check(lib().lwp_multiscale_accumulate(self.h.ptr, mp, mm, N, shp[-3], shp[-2], shp[-1], up_ratio, padv, H, W, n_scales, ap, am, 1 if init else 0), self.h.ptr)
        return accum

    def extract_keypoints(self, heatmap):
        """In-place threshold of ``heatmap`` (2-D float32 view) and key-point list [(x, y, score)]."""
        if not isinstance(heatmap, np.ndarray) or heatmap.ndim != 2 or heatmap.dtype != np.float32:
            raise TypeError("heatmap must be a 2-D float32 numpy array (view)")
        if not heatmap.flags.writeable:
            raise ValueError("heatmap must be writeable (it is thresholded in place)")
        H, W = heatmap.shape
        if heatmap.strides[0] % 4 or heatmap.strides[1] % 4:
            raise ValueError("unsupported strides")
        cap = self.caps[1]
        xs = np.empty(cap, np.int64); ys = np.empty(cap, np.int64); sc = np.empty(cap, np.float32)
        n = C.c_int()
        check(lib().lwp_extract_keypoints(self.h.ptr, heatmap.ctypes.data, H, W, heatmap.strides[0] // 4, heatmap.strides[1] // 4,
                                          xs.ctypes.data_as(C.POINTER(C.c_int64)), ys.ctypes.data_as(C.POINTER(C.c_int64)),
                                          sc.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)), self.h.ptr)
        return xs[:n.value], ys[:n.value], sc[:n.value]

    def group_keypoints(self, kpts, type_counts, pafs, demo):
        """kpts (K,4) float64, type_counts (18,), pafs (H,W,38) float32 -> (P,20) float64."""
        pafs = np.ascontiguousarray(pafs, dtype=np.float32)
        H, W, NPc = pafs.shape
        if NPc != self.NP:
            raise ValueError("pafs must have %d channels" % self.NP)
        kp = np.ascontiguousarray(kpts, dtype=np.float64).reshape(-1, 4)
        tc = np.ascontiguousarray(type_counts, dtype=np.int32)
        cap = self.caps[3]
        ent = np.empty((cap, 20), np.float64)
        n = C.c_int()
        check(lib().lwp_group_keypoints(self.h.ptr, kp.ctypes.data, tc.ctypes.data_as(C.POINTER(C.c_int)), pafs.ctypes.data,
                                        MEM_HOST, H, W, 1 if demo else 0, ent.ctypes.data, cap, C.byref(n)), self.h.ptr)
        return ent[:n.value].copy()

    # ------------------------------------------------------------------ fused pipeline
    def _result_buffers(self, N):
        kcap = 18 * self.caps[1]
        ecap = self.caps[3]
        return (np.zeros((N, 18), np.int32), np.zeros((N, kcap, 4), np.float64), np.zeros((N, ecap, 20), np.float64),
                np.zeros(N, np.int32), kcap, ecap)

    @staticmethod
    def _unpack(N, counts, kpts, ent, ne):
        out = []
        for f in range(N):
            K = int(counts[f].sum())
            out.append((ent[f, :ne[f]].copy(), kpts[f, :K].copy(), counts[f].copy()))
        return out

    def infer_poses(self, x, upsample_ratio=4, demo=True):
        """x: (N,3,H,W) float32 (numpy / cpu tensor / cuda tensor), already normalised and padded.
        Returns per frame (pose_entries (P,20) f64, all_keypoints (K,4) f64, type_counts (18,))."""
        torch = _torch()
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if isinstance(x, np.ndarray) else x
        t = t.detach().to(torch.float32).contiguous()
        N, _, H, W = t.shape
        counts, kpts, ent, ne, kcap, ecap = self._result_buffers(N)
        if t.is_cuda and getattr(x, "_lwp_stream_owner", None) is not self:
            self._order(t.device)
        check(lib().lwp_infer_poses(self.h.ptr, t.data_ptr(), MEM_DEVICE if t.is_cuda else MEM_HOST, N, H, W, upsample_ratio,
                                    1 if demo else 0, counts.ctypes.data_as(C.POINTER(C.c_int)), kpts.ctypes.data, kcap,
                                    ent.ctypes.data, ecap, ne.ctypes.data_as(C.POINTER(C.c_int))), self.h.ptr)
        return self._unpack(N, counts, kpts, ent, ne)

    def poses_from_maps(self, heat, paf, upsample_ratio=4, demo=True, layout="NCHW"):
        """Post-processing only.  layout "NCHW": the low-res stage outputs (N,19,h,w) / (N,38,h,w); layout "NHWC": full-res
        averaged maps (N,H,W,19) / (N,H,W,38) of the multi-scale path with upsample_ratio=1.  float32 numpy or cuda tensors."""
        lay = {"NCHW": 0, "NHWC": 1}[layout]
        if getattr(heat, "is_cuda", False):
            heat, paf = heat.detach().contiguous(), paf.detach().contiguous()
            self._order(heat.device)
            hp, pp, mem = heat.data_ptr(), paf.data_ptr(), MEM_DEVICE
        else:
            heat = np.ascontiguousarray(heat, dtype=np.float32)
            paf = np.ascontiguousarray(paf, dtype=np.float32)
            hp, pp, mem = heat.ctypes.data, paf.ctypes.data, MEM_HOST
        if lay == 0:
            N, _, hs, ws = heat.shape
        else:
            N, hs, ws, _ = heat.shape
        counts, kpts, ent, ne, kcap, ecap = self._result_buffers(N)
        check(lib().lwp_poses_from_maps(self.h.ptr, hp, pp, mem, lay, N, hs, ws, upsample_ratio,
                                        1 if demo else 0, counts.ctypes.data_as(C.POINTER(C.c_int)), kpts.ctypes.data, kcap,
                                        ent.ctypes.data, ecap, ne.ctypes.data_as(C.POINTER(C.c_int))), self.h.ptr)
        return self._unpack(N, counts, kpts, ent, ne)

    def layers(self):
        out = []
        name = C.create_string_buffer(128)
        v = [C.c_int() for _ in range(6)]
        macs = C.c_int64()
        for i in range(lib().lwp_layer_count(self.h.ptr)):
            check(lib().lwp_layer_info(self.h.ptr, i, name, 128, *[C.byref(x) for x in v], C.byref(macs)), self.h.ptr)
            out.append(dict(index=i, name=name.value.decode(), kind=v[0].value, cin=v[1].value, cout=v[2].value,
                            ksize=v[3].value, stride=v[4].value, dilation=v[5].value, macs_per_pixel=macs.value))
        return out

    def debug_layer_output(self, x, layer_index):
        """Output of layer ``layer_index`` (NCHW float32 numpy) for input x (N,3,H,W) numpy."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        N, _, H, W = x.shape
        info = self.layers()[layer_index]
        buf = np.empty(N * info["cout"] * ((H + 1) // 2) * ((W + 1) // 2), np.float32)
        dims = (C.c_int * 4)()
        check(lib().lwp_debug_layer_output(self.h.ptr, x.ctypes.data, N, H, W, layer_index, buf.ctypes.data, buf.size, dims), self.h.ptr)
        n = dims[0] * dims[1] * dims[2] * dims[3]
        return buf[:n].reshape(dims[0], dims[1], dims[2], dims[3]).copy()

    def frames_per_pass(self, N, H, W):
        """Frames one launch sequence of an (N,3,H,W) call takes (N unless the batch is split inside the call)."""
        n = lib().lwp_debug_frames_per_pass(self.h.ptr, N, H, W)
        if n < 0:
            raise ValueError("bad shape")
        return n

    def post_counts(self, frame=0):
        """(peaks per type before the NMS [18], key-points per type [18], scored connection candidates per limb [19], picked
        connections per limb [19]) of one frame of the last infer_poses / poses_from_maps call (debug)."""
        arrs = [(C.c_int * n)() for n in (18, 18, 19, 19)]
        check(lib().lwp_debug_post_counts(self.h.ptr, frame, *arrs), self.h.ptr)
        return tuple(np.array(list(a), dtype=np.int64) for a in arrs)

    def layer_variant(self, layer_index):
        """Kernel variant the last debug_layer_output / profile_launches pass picked for a layer ("" before any)."""
        name = C.create_string_buffer(96)
        check(lib().lwp_debug_layer_variant(self.h.ptr, layer_index, name, 96), self.h.ptr)
        return name.value.decode()

    def _as_device_input(self, x):
        """Checks shared by every entry point that hands ``data_ptr()`` of a resident frame batch to the library: float32,
        contiguous, (N,3,H,W), on this engine's GPU; work still queued on torch's current stream is ordered before the
        library's (own, non-blocking) stream by an event — the host does not wait."""
        torch = _torch()
        if not getattr(x, "is_cuda", False):
            raise TypeError("expected a cuda tensor")
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise TypeError("expected a float32 (N, 3, H, W) tensor, got %s %s" % (x.dtype, tuple(x.shape)))
        if not x.is_contiguous():
            raise ValueError("input tensor must be contiguous")
        if x.device.index != self.device_id:
            raise ValueError("input is on cuda:%d but the engine lives on cuda:%d" % (x.device.index, self.device_id))
        if getattr(x, "_lwp_stream_owner", None) is not self:     # (a tensor this engine produced on its own stream is in order already)
            self._order(x.device)
        return x

    def infer_poses_async(self, x_cuda, upsample_ratio=4, demo=True):
        x_cuda = self._as_device_input(x_cuda)
        N, _, H, W = x_cuda.shape
        self._keep = x_cuda
        check(lib().lwp_infer_poses_async(self.h.ptr, x_cuda.data_ptr(), N, H, W, upsample_ratio, 1 if demo else 0), self.h.ptr)
        self._last_N = N

    def fetch_poses(self):
        N = self._last_N
        counts, kpts, ent, ne, kcap, ecap = self._result_buffers(N)
        check(lib().lwp_fetch_poses(self.h.ptr, counts.ctypes.data_as(C.POINTER(C.c_int)), kpts.ctypes.data, kcap, ent.ctypes.data,
                                    ecap, ne.ctypes.data_as(C.POINTER(C.c_int))), self.h.ptr)
        return self._unpack(N, counts, kpts, ent, ne)

    # ------------------------------------------------------------------ pipelined streaming (two slots)
    def pipeline_submit(self, x_cuda, slot, upsample_ratio=4, demo=True):
        x_cuda = self._as_device_input(x_cuda)
        N, _, H, W = x_cuda.shape
        self._keep_slot = getattr(self, "_keep_slot", {})
        self._keep_slot[slot] = (x_cuda, N)
        check(lib().lwp_pipeline_submit(self.h.ptr, x_cuda.data_ptr(), N, H, W, upsample_ratio, 1 if demo else 0, slot), self.h.ptr)

    def pipeline_fetch(self, slot):
        _, N = self._keep_slot[slot]
        counts, kpts, ent, ne, kcap, ecap = self._result_buffers(N)
        check(lib().lwp_pipeline_fetch(self.h.ptr, slot, counts.ctypes.data_as(C.POINTER(C.c_int)), kpts.ctypes.data, kcap,
                                       ent.ctypes.data, ecap, ne.ctypes.data_as(C.POINTER(C.c_int))), self.h.ptr)
        return self._unpack(N, counts, kpts, ent, ne)

    # ------------------------------------------------------------------ measurement
    def time_pipeline(self, x_cuda, iters, what=1, upsample_ratio=4, demo=True):
        """milliseconds for ``iters`` back-to-back passes (HIP events on the engine's stream)."""
        x_cuda = self._as_device_input(x_cuda)
        N, _, H, W = x_cuda.shape
        ms = C.c_float()
        check(lib().lwp_time_pipeline(self.h.ptr, x_cuda.data_ptr(), N, H, W, upsample_ratio, 1 if demo else 0, what, iters, C.byref(ms)), self.h.ptr)
        self._last_N = N
        return ms.value

    def time_layer(self, layer_index, N, H, W, iters=50):
        ms = C.c_float()
        check(lib().lwp_debug_time_layer(self.h.ptr, layer_index, N, H, W, iters, C.byref(ms)), self.h.ptr)
        return ms.value

    def profile_launches(self, x_cuda, reps=5, upsample_ratio=4, demo=True):
        """[(name, class, ms)] per launch in issue order (HIP events around every launch)."""
        x_cuda = self._as_device_input(x_cuda)
        N, _, H, W = x_cuda.shape
        cap = 256
        ms = (C.c_float * cap)()
        kc = (C.c_int * cap)()
        n = C.c_int()
        check(lib().lwp_profile_launches(self.h.ptr, x_cuda.data_ptr(), N, H, W, upsample_ratio, 1 if demo else 0, reps, ms, kc, cap, C.byref(n)), self.h.ptr)
        layers = self.layers()
        post = ["find_peaks", "nms", "score_pairs", "match", "assemble"]
        out, npost = [], 0
        for i in range(n.value):
            li = (kc[i] >> 8) - 1                      # first layer the launch covers (a fused head pair is one launch), -1: post kernel
            if li >= 0:
                name = layers[li]["name"]
            else:
                name = post[npost] if npost < len(post) else "launch%d" % i
                npost += 1
            out.append((name, kc[i] & 0xff, ms[i]))
        return out

    def profile_classes(self, x_cuda, reps=5, upsample_ratio=4, demo=True):
        x_cuda = self._as_device_input(x_cuda)
        N, _, H, W = x_cuda.shape
        ms = (C.c_float * 6)()
        ln = (C.c_int * 6)()
        check(lib().lwp_profile_classes(self.h.ptr, x_cuda.data_ptr(), N, H, W, upsample_ratio, 1 if demo else 0, reps, ms, ln), self.h.ptr)
        names = ["stem", "depthwise", "pointwise_1x1", "dense_3x3", "post", "other"]
        return {n: {"ms": ms[i], "launches": ln[i]} for i, n in enumerate(names)}


_default = {}


def default_engine(device_id=None):
    """Shared engine for the free functions (extract_keypoints / group_keypoints / up-sampling)."""
    if device_id is None:
        device_id = 0
    if device_id not in _default:
        _default[device_id] = Engine(device_id)
    return _default[device_id]
