"""One-shot all-reduce of the packed statistics over IPC-mapped mailboxes (csrc/ipc_allreduce.hip): the opt-in,
latency-minimal alternative to the RCCL all-reduce of svae_amd/parallel.py for the ONE collective of a training step
(/root/reference/svae/svae.py:33-34 consumes the batch-summed statistics; 4n^2+n+2 doubles).

    ar = MailboxAllReduce(n_doubles, group)      # once: fine-grained mailbox, IPC handles exchanged through `group`
    ar(packed)                                   # in place, asynchronous on the current stream; every rank gets the same bits
    ar.check()                                   # (also every `check_every` calls) raises if a peer never published

`n_doubles` is the mailbox CAPACITY: it fixes the mailbox layout for its lifetime; a call may reduce any shorter
buffer, and consecutive calls may differ in length.  A peer that does not show up within the spin limit (minutes by
default) does not pass for a sum: the elements come out as NaN, the status word is raised, and the next check -- the
caller's, or the automatic one every `check_every` calls -- raises.

One process per GPU of ONE node (or several processes on one GPU: the tests).  The mailbox is fine-grained device
memory (hipExtMallocWithFlags: remote stores must be visible to the owner's polling loads while its kernel runs, which
coarse-grained memory does not promise); peers map it with hipIpcOpenMemHandle.  `HSA_ENABLE_IPC_MODE_LEGACY=0` must be
set (dmabuf IPC), as for RCCL.  The handles travel through torch.distributed (any backend) once, at construction.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib

_HIP_DEVICE_MALLOC_FINEGRAINED = 0x1
_HIP_IPC_MEM_LAZY_ENABLE_PEER_ACCESS = 0x1
_hip = None


class _IpcHandle(ctypes.Structure):          # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
    _fields_ = [("reserved", ctypes.c_char * 64)]


def _runtime_path():
    """path of the libamdhip64 this process has ALREADY mapped (torch's): loading by a versioned soname would fail on
    another ROCm release and could map a second runtime next to torch's"""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64.so" in path:
                    return path
    except OSError:
        pass
    return "libamdhip64.so"


def _runtime():
    """the HIP runtime torch already mapped (one runtime per process)"""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL(_runtime_path())
        _hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        _hip.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_IpcHandle), ctypes.c_void_p]
        _hip.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _IpcHandle, ctypes.c_uint]
        _hip.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
    return _hip


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


class MailboxAllReduce(object):
    def __init__(self, n_doubles, group=None, device=None, spin_limit=0, check_every=64):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("MailboxAllReduce needs an initialised torch.distributed process group")
        self.group = group
        self.n = int(n_doubles)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lib = _lib.load()
        nbytes = int(self.lib.svae_ipc_mailbox_bytes(self.n, self.world))
        if nbytes == 0:
            raise ValueError("mailbox all-reduce: 1 <= world <= 16, n >= 1")
        hip = _runtime()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            self._mine = ctypes.c_void_p()
            _check(hip.hipExtMallocWithFlags(ctypes.byref(self._mine), nbytes, _HIP_DEVICE_MALLOC_FINEGRAINED),
                   "hipExtMallocWithFlags(fine-grained)")
            _check(hip.hipMemset(self._mine, 0, nbytes), "hipMemset")
            handle = _IpcHandle()
            _check(hip.hipIpcGetMemHandle(ctypes.byref(handle), self._mine), "hipIpcGetMemHandle")
            torch.cuda.synchronize()
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(bytearray(handle)), group=group)   # (also a barrier: every mailbox is zeroed)
            self._opened = []
            ptrs = []
            for q, raw in enumerate(handles):
                if q == self.rank:
                    ptrs.append(self._mine.value)
                    continue
                h = _IpcHandle.from_buffer_copy(raw)
                pp = ctypes.c_void_p()
                _check(hip.hipIpcOpenMemHandle(ctypes.byref(pp), h, _HIP_IPC_MEM_LAZY_ENABLE_PEER_ACCESS),
                       "hipIpcOpenMemHandle (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)")
                self._opened.append(pp)
                ptrs.append(pp.value)
        self._boxes = (ctypes.c_void_p * self.world)(*ptrs)
        self.info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.epoch = 0
        self.spin_limit = int(spin_limit)         # polls per word before a peer counts as absent (0: the library's 2^28)
        self.check_every = int(check_every)       # every so many calls the status word is read (one host sync); 0: never

    def __call__(self, packed):
        """in-place sum over the ranks of `packed` (float64, contiguous, <= n doubles), on the current stream"""
        if packed.dtype != torch.float64 or not packed.is_contiguous() or packed.numel() > self.n:
            raise ValueError("mailbox all-reduce: contiguous float64 buffer of at most %d elements" % self.n)
        self.epoch += 1
        if self.epoch >= 1 << 31:
            raise OverflowError("mailbox all-reduce: epoch counter exhausted")
        # (the mailbox is laid out with its CAPACITY self.n whatever the length of this call: see csrc/ipc_allreduce.hip)
        rc = self.lib.svae_ipc_allreduce_f64(packed.numel(), self.n, self.rank, self.world, self.epoch, self.spin_limit,
                                             _lib.ptr(packed), _lib.ptr(packed),
                                             ctypes.cast(self._boxes, ctypes.c_void_p), _lib.ptr(self.info),
                                             _lib.current_stream(self.device))
        _lib.check(rc, "svae_ipc_allreduce_f64")
        if self.check_every and self.epoch % self.check_every == 0 \
                and not torch.cuda.is_current_stream_capturing():
            # (a blocking read of the status word: skipped under hipGraph capture, where it is illegal -- a captured
            #  loop checks through `check()` between replays; pass check_every=0 for fully asynchronous pipelines)
            self.check()
        return packed

    def check(self):
        """host synchronisation: raises if a call timed out waiting for a peer"""
        v = int(self.info.item())
        if v != 0:
            raise RuntimeError("mailbox all-reduce: a peer never published (info %d); the affected elements are NaN" % v)

    def close(self):
        hip = _runtime()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier(self.group)                # nobody is still storing into a mailbox that is about to go
        for pp in getattr(self, "_opened", []):
            hip.hipIpcCloseMemHandle(pp)
        self._opened = []
        if getattr(self, "_mine", None) is not None and self._mine.value:
            hip.hipFree(self._mine)
            self._mine = ctypes.c_void_p()
