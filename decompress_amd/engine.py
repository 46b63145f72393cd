"""Batch engine: device buffers in, device buffers out (torch is only the
allocator / stream provider here; the work is done by libmdeflate.so)."""
import ctypes

import numpy as _np

from . import _lib

FORMAT_DEFLATE = 0
FORMAT_ZLIB = 1
FORMAT_GZIP = 2
MATCHER_DE, MATCHER_LZ = 0, 1
DRIVER_ZL, DRIVER_HIGHER, DRIVER_CLI = 0, 1, 2

# variant names of De.Inf.Ns.error (lib/de.ml:1548-1555) + Zl.Inf.Ns.error (lib/zl.ml:383)
STATUS_NAMES = {
    0: "Ok", 1: "Unexpected_end_of_input", 2: "Unexpected_end_of_output",
    3: "Invalid_kind_of_block", 4: "Invalid_dictionary",
    5: "Invalid_complement_of_length", 6: "Invalid_distance",
    7: "Invalid_distance_code", 8: "Invalid_header", 9: "Invalid_checksum",
    10: "Invalid GZip header", 11: "Invalid GZip header checksum", 12: "Invalid input size",
    13: "Queue.Full", 14: "Invalid input", 15: "No dictionary at offset 0 available",
    16: "Input is malformed or output is not large enough",
}
STATUS_CODES = {v: k for k, v in STATUS_NAMES.items()}


class Error(RuntimeError):
    """Call-level failure (the reference raises Invalid_argument / Failure)."""


class _HostArray(_np.ndarray):
    """numpy view of a pinned block of md_host_alloc; carries the owner that frees it"""

    def __array_finalize__(self, obj):
        self._md_owner = getattr(obj, "_md_owner", None)


def _check_blob(a, name):
    """The host entry points take the blob's address and byte count: it has to be a C-contiguous uint8 array (a strided
    view or another dtype would hand the library a wrong extent)."""
    if not isinstance(a, _np.ndarray) or a.dtype != _np.uint8 or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("%s must be a C-contiguous numpy uint8 array" % name)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Engine:
    """One engine per GPU (wraps md_ctx).  Kernels are enqueued on torch's current stream of `device` — the
    legacy default stream included (MD_STREAM_NULL), so batch calls are ordered with the torch kernels that
    produce their inputs and consume their results — unless own_stream=True (then the caller synchronises)."""

    def __init__(self, device=0, own_stream=False):
        import torch

        self.torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise Error("decompress_amd needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.own_stream = bool(own_stream)
        if own_stream:
            handle = None
        else:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            handle = ctypes.c_void_p(stream) if stream else ctypes.c_void_p(_lib.MD_STREAM_NULL)
        self.ctx = self.lib.md_create(device, handle)
        if not self.ctx:
            raise Error(self.lib.md_last_error_string(None).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.md_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise Error("%s: %s" % (self.lib.md_status_string(rc).decode(),
                                    self.lib.md_last_error_string(self.ctx).decode()))

    def lzo_batch(self, compress, d_in, in_off, in_len, d_out, out_off, out_cap, results=None):
        """Lzo.compress / Lzo.uncompress over n device-resident buffers -> (out_len, status) CUDA tensors (async)."""
        torch = self.torch
        n = in_off.numel()
        if results is None:
            results = (torch.empty(n, dtype=torch.int64, device=self.device),
                       torch.empty(n, dtype=torch.int32, device=self.device))
        fn = self.lib.md_lzo_compress_batch_device if compress else self.lib.md_lzo_uncompress_batch_device
        self._check(fn(self.ctx, n, _ptr(d_in), _ptr(in_off), _ptr(in_len), _ptr(d_out), _ptr(out_off), _ptr(out_cap),
                       _ptr(results[0]), _ptr(results[1])))
        return results

    def lzo_many(self, compress, bufs, caps):
        """Convenience for tests: list of bytes -> list of (status, bytes)."""
        import numpy as np

        torch = self.torch
        n = len(bufs)
        if n == 0:
            return []
        in_len = np.array([len(s) for s in bufs], dtype=np.int64)
        in_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((in_len + 31) // 16 * 16)[:-1], out=in_off[1:])
        cap = np.array(caps, dtype=np.int64)
        out_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((cap + 255) // 256 * 256)[:-1], out=out_off[1:])
        blob = np.zeros(int(in_off[-1] + in_len[-1]) + 32, dtype=np.uint8)
        for s, o in zip(bufs, in_off):
            blob[o:o + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
        dev = self.device
        t = lambda a: torch.from_numpy(a).to(dev)
        d_out = torch.zeros(int(out_off[-1] + cap[-1]) + 32, dtype=torch.uint8, device=dev)
        out_len, status = self.lzo_batch(compress, t(blob), t(in_off), t(in_len), d_out, t(out_off), t(cap))
        torch.cuda.synchronize(dev)
        out = d_out.cpu().numpy()
        out_len, status = out_len.cpu().numpy(), status.cpu().numpy()
        return [(int(status[i]), out[out_off[i]:out_off[i] + out_len[i]].tobytes()) for i in range(n)]

    def set_matcher(self, matcher):
        """Default matcher of this Engine object's later deflate calls: 0 = De.Lz77, 1 = Lz (lib/lz.ml).  Kept on the
        Python side — the C ABI takes the matcher per call (md_deflate_params)."""
        if matcher not in (MATCHER_DE, MATCHER_LZ):
            raise Error("Invalid argument: unknown matcher")
        self._matcher = int(matcher)

    def gz_set_header(self, mtime=0, os=3, hcrc=False, ascii=False, filename=None, comment=None):
        """Default GZip header of this Engine object's later FORMAT_GZIP deflate calls (Gz.Def.encoder's fields,
        lib/gz.ml:859-918); passed per call through md_deflate_params.gz_header."""
        self._gz = dict(mtime=int(mtime) & 0xffffffff, os=int(os), hcrc=int(bool(hcrc)), ascii=int(bool(ascii)),
                        filename=filename, comment=comment)

    def _params(self, level, queue, driver, dynamic, matcher=None, header=None, total_in=0):
        h = header if header is not None else getattr(self, "_gz", None)
        gz = _lib.GzHeader(**h) if h else None
        p = _lib.DeflateParams(int(level), int(queue), int(driver), int(bool(dynamic)),
                               int(getattr(self, "_matcher", MATCHER_DE) if matcher is None else matcher),
                               ctypes.pointer(gz) if gz is not None else None, 0, int(total_in))
        p._keep = gz  # the struct must outlive the call
        return p

    def crc32_batch(self, d_data, d_off, d_len):
        """Checkseum.Crc32 of n device-resident buffers -> uint32 tensor (device)."""
        n = d_off.numel()
        crc = self.torch.empty(n, dtype=self.torch.int32, device=self.device)
        self._check(self.lib.md_crc32_batch_device(self.ctx, n, _ptr(d_data), _ptr(d_off), _ptr(d_len), _ptr(crc)))
        return crc

    def set_option(self, key, value):
        self._check(self.lib.md_set_option(self.ctx, key.encode(), int(value)))

    PROFILE_FIELDS = ["cyc_ensure", "cyc_decode1", "cyc_decode2", "cyc_emit_a", "cyc_far", "cyc_near",
                      "cyc_adler", "cyc_header", "cyc_near_fast", "cyc_near_slow", "cyc_near_upd", "cyc_near_load", "cyc_hdr_lens", "cyc_hdr_lit", "cyc_far_rec", "cyc_far_load", "cyc_wait_dec", "cyc_wait_copy", "rounds", "passes", "lanes", "tokens", "slots",
                      "near_iters", "end_chain", "end_fit", "end_records", "end_stage", "end_eob", "long_near"]

    def get_profile(self):
        """In-kernel phase profile of stream 0 (enable with set_option('profile', 1))."""
        buf = (ctypes.c_uint64 * 32)()
        self._check(self.lib.md_get_profile(self.ctx, buf))
        return dict(zip(self.PROFILE_FIELDS, list(buf)))

    def get_profile_raw(self):
        """The 32 raw profile words (the deflate kernel's layout differs from the inflate one)."""
        buf = (ctypes.c_uint64 * 32)()
        self._check(self.lib.md_get_profile(self.ctx, buf))
        return list(buf)

    def synchronize(self):
        self._check(self.lib.md_synchronize(self.ctx))

    def timing_begin(self):
        self._check(self.lib.md_timing_begin(self.ctx))

    def timing_end(self):
        ms = ctypes.c_float()
        self._check(self.lib.md_timing_end(self.ctx, ctypes.byref(ms)))
        return ms.value

    # ------------------------------------------------------------------ inflate
    def inflate_batch(self, fmt, d_in, in_off, in_len, d_out, out_off, out_cap, results=None):
        """All arguments are CUDA tensors: d_in/d_out uint8, descriptors int64.
        Returns (out_len, consumed, status, checksum) CUDA tensors (async)."""
        torch = self.torch
        n = in_off.numel()
        if results is None:
            results = (torch.empty(n, dtype=torch.int64, device=self.device),
                       torch.empty(n, dtype=torch.int64, device=self.device),
                       torch.empty(n, dtype=torch.int32, device=self.device),
                       torch.empty(n, dtype=torch.int32, device=self.device))
        out_len, consumed, status, checksum = results
        self._check(self.lib.md_inflate_batch_device(
            self.ctx, fmt, n, _ptr(d_in), _ptr(in_off), _ptr(in_len), _ptr(d_out), _ptr(out_off),
            _ptr(out_cap), _ptr(out_len), _ptr(consumed), _ptr(status), _ptr(checksum)))
        return results

    # ------------------------------------------------------------------ host buffers (md_*_batch_host)
    def host_buffer(self, nbytes):
        """A pinned host buffer of nbytes (md_host_alloc) as a numpy uint8 array; the array owns it (freed with it)."""
        import numpy as np

        p = self.lib.md_host_alloc(self.ctx, nbytes)
        if not p:
            raise MemoryError("md_host_alloc(%d)" % nbytes)
        lib, ctx = self.lib, self.ctx

        class _Owner:
            def __init__(self):
                self.buf = (ctypes.c_uint8 * max(1, nbytes)).from_address(p)

            def __del__(self):
                lib.md_host_free(None, p)  # (the context may be gone by now: md_host_free does not use it)

        owner = _Owner()
        a = np.frombuffer(owner.buf, dtype=np.uint8, count=nbytes)
        a = a.view(_HostArray)
        a._md_owner = owner  # the array (and every view of it) keeps the pinned block alive; it goes when they do
        return a

    def inflate_batch_host(self, fmt, h_in, in_off, in_len, h_out, out_off, out_cap):
        """md_inflate_batch_host: numpy uint8 blobs (pinned ones overlap copies and kernels) and uint64 descriptor arrays.
        Returns (out_len, consumed, status, checksum) numpy arrays.  Synchronous."""
        import numpy as np

        n = len(in_off)
        _check_blob(h_in, "h_in"), _check_blob(h_out, "h_out")
        u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        in_off, in_len, out_off, out_cap = u64(in_off), u64(in_len), u64(out_off), u64(out_cap)
        out_len, consumed = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        status, checksum = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.uint32)
        self._check(self.lib.md_inflate_batch_host(
            self.ctx, fmt, n, h_in.ctypes.data, h_in.nbytes, in_off.ctypes.data, in_len.ctypes.data, h_out.ctypes.data,
            h_out.nbytes, out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data, consumed.ctypes.data,
            status.ctypes.data, checksum.ctypes.data))
        return out_len, consumed, status, checksum

    def deflate_batch_host(self, fmt, h_in, in_off, in_len, h_out, out_off, out_cap, level=6, queue=4096, driver=DRIVER_ZL,
                           dynamic=True, matcher=None, header=None):
        """md_deflate_batch_host, as inflate_batch_host.  Returns (out_len, status, checksum of the input)."""
        import numpy as np

        n = len(in_off)
        _check_blob(h_in, "h_in"), _check_blob(h_out, "h_out")
        u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        in_off, in_len, out_off, out_cap = u64(in_off), u64(in_len), u64(out_off), u64(out_cap)
        out_len = np.zeros(n, dtype=np.uint64)
        status, checksum = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.uint32)
        params = self._params(level, queue, driver, dynamic, matcher, header, 0)
        self._check(self.lib.md_deflate_batch_host(
            self.ctx, fmt, ctypes.byref(params), n, h_in.ctypes.data, h_in.nbytes, in_off.ctypes.data, in_len.ctypes.data,
            h_out.ctypes.data, h_out.nbytes, out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data,
            status.ctypes.data, checksum.ctypes.data))
        return out_len, status, checksum

    def inflate_many(self, streams, caps, fmt=FORMAT_DEFLATE):
        """Convenience for tests: list of bytes -> list of (status, consumed, bytes, adler)."""
        import numpy as np

        torch = self.torch
        n = len(streams)
        if n == 0:
            return []
        in_len = np.array([len(s) for s in streams], dtype=np.int64)
        in_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((in_len + 15) // 16 * 16)[:-1], out=in_off[1:])
        cap = np.array(caps, dtype=np.int64)
        out_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((cap + 255) // 256 * 256)[:-1], out=out_off[1:])
        blob = np.zeros(int(in_off[-1] + in_len[-1]) + 16, dtype=np.uint8)
        for s, o in zip(streams, in_off):
            blob[o:o + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
        dev = self.device
        d_in = torch.from_numpy(blob).to(dev)
        d_out = torch.zeros(int(out_off[-1] + cap[-1]) + 16, dtype=torch.uint8, device=dev)
        t = lambda a: torch.from_numpy(a).to(dev)
        args = (fmt, d_in, t(in_off), t(in_len), d_out, t(out_off), t(cap))
        if self.own_stream:  # torch's fills and copies above ran on ITS stream: they come first (the caller synchronises, i.e. we)
            torch.cuda.synchronize(dev)
        out_len, consumed, status, checksum = self.inflate_batch(*args)
        torch.cuda.synchronize(dev)
        out = d_out.cpu().numpy()
        out_len, consumed, status = out_len.cpu().numpy(), consumed.cpu().numpy(), status.cpu().numpy()
        checksum = checksum.cpu().numpy().view(np.uint32)
        return [(int(status[i]), int(consumed[i]),
                 out[out_off[i]:out_off[i] + out_len[i]].tobytes(), int(checksum[i]))
                for i in range(n)]


    # ------------------------------------------------------------------ deflate
    def deflate_batch(self, fmt, d_in, in_off, in_len, d_out, out_off, out_cap, level=6, queue=4096,
                      driver=DRIVER_ZL, dynamic=True, results=None, matcher=None, header=None, total_in=0):
        """All arguments are CUDA tensors (uint8 data, int64 descriptors).
        Returns (out_len, status, adler32_of_input) CUDA tensors (async).  total_in: an upper bound of in_len.sum()
        when the caller knows it (md_deflate_params.total_in_bytes); 0 makes the engine read the sum back first."""
        torch = self.torch
        n = in_off.numel()
        if results is None:
            results = (torch.empty(n, dtype=torch.int64, device=self.device),
                       torch.empty(n, dtype=torch.int32, device=self.device),
                       torch.empty(n, dtype=torch.int32, device=self.device))
        out_len, status, checksum = results
        params = self._params(level, queue, driver, dynamic, matcher, header, total_in)
        self._check(self.lib.md_deflate_batch_device(
            self.ctx, fmt, ctypes.byref(params), n, _ptr(d_in), _ptr(in_off), _ptr(in_len),
            _ptr(d_out), _ptr(out_off), _ptr(out_cap), _ptr(out_len), _ptr(status), _ptr(checksum)))
        return results

    def deflate_many(self, bufs, fmt=FORMAT_DEFLATE, level=6, queue=4096, driver=DRIVER_ZL, dynamic=True,
                     caps=None, matcher=None, header=None):
        """Convenience for tests: list of bytes -> list of (status, compressed bytes, adler32 of input)."""
        import numpy as np

        torch = self.torch
        n = len(bufs)
        if n == 0:
            return []
        in_len = np.array([len(s) for s in bufs], dtype=np.int64)
        in_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((in_len + 15) // 16 * 16)[:-1], out=in_off[1:])
        if caps is None:
            caps = [2 * len(s) + 8192 for s in bufs]
        cap = np.array(caps, dtype=np.int64)
        out_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((cap + 255) // 256 * 256)[:-1], out=out_off[1:])
        blob = np.zeros(int(in_off[-1] + in_len[-1]) + 16, dtype=np.uint8)
        for s, o in zip(bufs, in_off):
            blob[o:o + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
        dev = self.device
        d_in = torch.from_numpy(blob).to(dev)
        d_out = torch.zeros(int(out_off[-1] + cap[-1]) + 16, dtype=torch.uint8, device=dev)
        t = lambda a: torch.from_numpy(a).to(dev)
        d_off, d_len, d_ooff, d_cap = t(in_off), t(in_len), t(out_off), t(cap)
        if self.own_stream:
            torch.cuda.synchronize(dev)
        out_len, status, checksum = self.deflate_batch(fmt, d_in, d_off, d_len, d_out, d_ooff, d_cap,
                                                       level, queue, driver, dynamic, matcher=matcher, header=header,
                                                       total_in=max(1, int(in_len.sum())))
        torch.cuda.synchronize(dev)
        out = d_out.cpu().numpy()
        out_len, status = out_len.cpu().numpy(), status.cpu().numpy()
        checksum = checksum.cpu().numpy().view(np.uint32)
        return [(int(status[i]), out[out_off[i]:out_off[i] + out_len[i]].tobytes(), int(checksum[i]))
                for i in range(n)]


    # ------------------------------------------------------------------ De.Def.Ns
    def def_ns_many(self, bufs, level=4, fmt=FORMAT_DEFLATE, caps=None):
        """list of bytes -> list of (status, compressed bytes, adler32 of input) through md_def_ns_batch_device"""
        import numpy as np

        torch = self.torch
        n = len(bufs)
        if n == 0:
            return []
        in_len = np.array([len(s) for s in bufs], dtype=np.int64)
        in_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((in_len + 15) // 16 * 16)[:-1], out=in_off[1:])
        if caps is None:
            caps = [int(self.lib.md_zl_def_ns_compress_bound(len(s))) for s in bufs]
        cap = np.array(caps, dtype=np.int64)
        out_off = np.zeros(n, dtype=np.int64)
        np.cumsum(((cap + 255) // 256 * 256)[:-1], out=out_off[1:])
        blob = np.zeros(int(in_off[-1] + in_len[-1]) + 16, dtype=np.uint8)
        for s, o in zip(bufs, in_off):
            blob[o:o + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
        dev = self.device
        d_in = torch.from_numpy(blob).to(dev)
        d_out = torch.zeros(int(out_off[-1] + cap[-1]) + 16, dtype=torch.uint8, device=dev)
        t = lambda a: torch.from_numpy(a).to(dev)
        out_len = torch.empty(n, dtype=torch.int64, device=dev)
        status = torch.empty(n, dtype=torch.int32, device=dev)
        checksum = torch.empty(n, dtype=torch.int32, device=dev)
        d_ioff, d_ilen, d_ooff, d_cap = t(in_off), t(in_len), t(out_off), t(cap)
        self._check(self.lib.md_def_ns_batch_device(self.ctx, fmt, int(level), max(1, int(in_len.sum())), n, _ptr(d_in), _ptr(d_ioff),
                                                    _ptr(d_ilen), _ptr(d_out), _ptr(d_ooff), _ptr(d_cap), _ptr(out_len),
                                                    _ptr(status), _ptr(checksum)))
        torch.cuda.synchronize(dev)
        out = d_out.cpu().numpy()
        out_len, status = out_len.cpu().numpy(), status.cpu().numpy()
        checksum = checksum.cpu().numpy().view(np.uint32)
        return [(int(status[i]), out[out_off[i]:out_off[i] + out_len[i]].tobytes(), int(checksum[i])) for i in range(n)]


_default = {}


def default_engine(device=0):
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]
