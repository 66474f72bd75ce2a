"""One decode/encode context per host thread and HIP stream (like one NativeWriter or one
column reader per thread upstream: src/read/deserialize.rs:28, iterators are Send + Sync and
share no state)."""
import ctypes as C

from . import _native as N


class Context:
    def __init__(self, device=0, stream=None):
        """`stream`: a torch.cuda.Stream to enqueue on (default: a new stream owned by the context;
        the legacy null stream cannot be passed through the C ABI, its handle is NULL)."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("strawboat_amd needs a GPU: torch.cuda.is_available() is False "
                               "(there is no CPU fallback)")
        self._lib = N.load()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.torch_device)
            if stream.cuda_stream == 0:
                stream = torch.cuda.Stream(device=self.torch_device)
        self.torch_stream = stream
        h = C.c_void_p()
        rc = self._lib.sb_ctx_create(self.device, C.c_void_p(self.torch_stream.cuda_stream), C.byref(h))
        if rc != N.SB_OK:
            raise N.NativeError(rc, "sb_ctx_create failed")
        self._h = h
        self._keep = []  # objects that must outlive the enqueued work

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sb_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, drain=True):
        if rc != N.SB_OK:
            msg = self._lib.sb_ctx_last_error(self._h)
            text = msg.decode() if msg else ""
            if drain:  # an enqueue call failed: drop the context's pending state with it
                self._lib.sb_ctx_synchronize(self._h)
                self._keep = []
            raise N.NativeError(rc, text)

    def synchronize(self):
        """Waits for the enqueued work; raises the first error a kernel reported."""
        rc = self._lib.sb_ctx_synchronize(self._h)
        keep, self._keep = self._keep, []
        try:
            self._check(rc, drain=False)
        finally:
            del keep

    def profile(self, enable=True):
        """Per-kernel HIP-event timing on this context's stream (resets the totals)."""
        self._check(self._lib.sb_ctx_profile(self._h, 1 if enable else 0))

    def zstd_block_stats(self):
        """(frames decoded block-parallel, frames handed back to the frame-serial decoder, blocks, sequences) so far"""
        out = (C.c_uint64 * 4)()
        self._check(self._lib.sb_ctx_zstd_block_stats(self._h, out), drain=False)
        return tuple(int(x) for x in out)

    def side_forks(self):
        """calls so far whose kernels were spread over side streams next to this context's stream (diagnostics)"""
        return int(self._lib.sb_ctx_side_forks(self._h))

    def replays(self):
        """synchronize intervals so far that were issued a second time because a kernel skipped on a hint was needed after all"""
        return int(self._lib.sb_ctx_replays(self._h))

    def profile_read(self):
        """{kernel name: (launches, total_ms)} accumulated up to the last synchronize()."""
        arr = (N.KernelStatC * 96)()
        n = self._lib.sb_ctx_profile_read(self._h, arr, 96)
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(n)}
