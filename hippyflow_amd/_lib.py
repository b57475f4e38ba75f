"""ctypes binding of libhfmi.so (include/hfmi.h).  No CPU fallback: every compute
entry point raises if the library or a GPU is missing."""
import ctypes as C
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HFMI_LIB") or os.path.join(HERE, "libhfmi.so")   # HFMI_LIB: A/B builds (scripts/)

LAYOUT_VECTORS = 0
LAYOUT_DENSE = 1
QR_CHOL, QR_MGS, QR_AUTO = 0, 1, 2
UNIQUE_ID_BYTES = 256
REDUCE_SUM, REDUCE_AVG, REDUCE_MAX = 0, 1, 2

HOST_APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_int)
POST_APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class HfmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libhfmi error %d: %s" % (code, message))
        self.code = code


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_D = C.POINTER(C.c_double)

# name -> argtypes; every function returns int status except the two noted below
SIGNATURES = {
    "hfmi_device_count": [C.POINTER(C.c_int)],
    "hfmi_ctx_create": [C.c_int, _PP],
    "hfmi_ctx_destroy": [_P],
    "hfmi_ctx_set_stream": [_P, _P],
    "hfmi_ctx_get_stream": [_P, _PP],
    "hfmi_ctx_synchronize": [_P],
    "hfmi_ctx_device_info": [_P, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)],
    "hfmi_ctx_pci_bus_id": [_P, C.c_char_p, C.c_int],
    "hfmi_timer_start": [_P],
    "hfmi_timer_stop": [_P, _D],
    "hfmi_block_create": [_P, C.c_int64, C.c_int, _PP],
    "hfmi_block_wrap": [_P, _P, C.c_int64, C.c_int, C.c_int64, _PP],
    "hfmi_block_view": [_P, C.c_int, C.c_int, _PP],
    "hfmi_block_destroy": [_P],
    "hfmi_block_info": [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64), _PP],
    "hfmi_block_upload": [_P, _P, C.c_int],
    "hfmi_block_download": [_P, _P, C.c_int],
    "hfmi_host_alloc_pinned": [C.c_size_t, C.POINTER(C.c_void_p)],
    "hfmi_host_free_pinned": [C.c_void_p],
    "hfmi_block_upload_async": [_P, _P, C.c_int, C.POINTER(C.c_int64)],
    "hfmi_ingest_wait": [_P, C.c_int64],
    "hfmi_ingest_fence": [_P],
    "hfmi_block_zero": [_P],
    "hfmi_block_copy": [_P, _P],
    "hfmi_block_scale": [_P, C.c_double],
    "hfmi_block_axpy": [_P, C.c_double, _P],
    "hfmi_block_norms": [_P, _P],
    "hfmi_randn_fill": [_P, C.c_uint64, C.c_uint32, C.c_double],
    "hfmi_philox_raw": [_P, C.c_uint64, C.c_uint32, _P],
    "hfmi_block_fill_matern32": [_P, C.c_int, C.c_int, C.c_double, C.c_double],
    "hfmi_block_dot": [_P, _P, _P],
    "hfmi_block_gemm_small": [_P, _P, C.c_double, C.c_double, _P],
    "hfmi_csr_create": [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _PP],
    "hfmi_csr_destroy": [_P],
    "hfmi_op_snapshot_gram": [_P, _P, C.c_double, _PP],
    "hfmi_op_low_rank": [_P, _P, _P, _PP],
    "hfmi_op_jtj": [_P, _P, C.c_int, C.c_int, _P, C.c_double, _PP],
    "hfmi_op_jjt": [_P, _P, C.c_int, C.c_int, C.c_double, _PP],
    "hfmi_op_dense_sym": [_P, _P, _PP],
    "hfmi_op_csr": [_P, _P, _PP],
    "hfmi_op_csr_pcg": [_P, _P, C.c_double, C.c_int, _PP],
    "hfmi_op_solver_info": [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "hfmi_op_compose3": [_P, _P, _P, _P, _PP],
    "hfmi_op_host_callback": [_P, HOST_APPLY_FN, _P, C.c_int64, _PP],
    "hfmi_op_host_set_chunk": [_P, C.c_int],
    "hfmi_op_set_post_apply": [_P, POST_APPLY_FN, _P],
    "hfmi_op_set_collective": [_P, _P, C.c_int],
    "hfmi_op_apply": [_P, _P, _P, C.c_int],
    "hfmi_comm_unique_id": [_P],
    "hfmi_comm_init_rank": [_P, _P, C.c_int, C.c_int, _PP],
    "hfmi_comm_init_from_file": [_P, C.c_char_p, C.c_int, C.c_int, _PP],
    "hfmi_comm_info": [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "hfmi_comm_describe": [_P, C.c_char_p, C.c_int],
    "hfmi_comm_decide_transport": [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.c_int,
                                   C.POINTER(C.c_int), C.c_char_p, C.c_int],
    "hfmi_comm_barrier": [_P],
    "hfmi_allreduce": [_P, _P, C.c_int],
    "hfmi_bcast": [_P, _P, C.c_int],
    "hfmi_allreduce_host": [_P, _P, C.c_int64, C.c_int],
    "hfmi_bcast_host": [_P, _P, C.c_int64, C.c_int],
    "hfmi_comm_destroy": [_P],
    "hfmi_op_destroy": [_P],
    "hfmi_borth_qr": [_P, _P, _P, _P, C.c_int, C.POINTER(C.c_int)],
    "hfmi_sym_eig_small": [_P, _P, C.c_int, C.c_int, _P, _P],
    "hfmi_sym_eig_leading": [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    "hfmi_block_gram_eig": [_P, _P, C.c_int, C.c_int, _P, _P],
    "hfmi_svd_small": [_P, _P, C.c_int, _P, _P, _P],
    "hfmi_double_pass": [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    "hfmi_double_pass_g": [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    "hfmi_bench_tsgemm_tn": [_P, _P, C.c_int, C.c_int, _P, _D],
    "hfmi_bench_tsgemm_nn": [_P, _P, _P, C.c_int, _D],
    "hfmi_bench_peaks": [_P, _D, _D, _D],
    "hfmi_bench_loaded_peak": [_P, _D, _D],
    "hfmi_bench_random_peaks": [_P, _D, _D, _D],
    "hfmi_bench_hbm_read": [_P, _D],
    "hfmi_dense_matmul": [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _D],
    "hfmi_bench_dgemm": [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _D, _D],
    "hfmi_profile_begin": [_P],
    "hfmi_profile_phases": [_P, _D],
    "hfmi_tuning_set": [C.c_char_p, C.c_int],
    "hfmi_profile_end": [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), _D,
                         C.POINTER(C.c_int64), _D, _D],
}
NON_STATUS = {"hfmi_last_error": (C.c_char_p, []), "hfmi_version": (C.c_int, []), "hfmi_build_tag": (C.c_char_p, [])}

_lib = None


def _preload_hip_runtime():
    """If PyTorch is installed, bind to ITS HIP runtime so that a later `import torch` (used only for
    torch.distributed / RCCL plumbing) shares one runtime with libhfmi instead of loading a second copy."""
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            return
        # the communicator opens librccl at run time: pair torch's HIP runtime with torch's RCCL build (and let a later
        # `import torch` find that copy already loaded) unless the caller chose one
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("HFMI_RCCL_LIB", rccl)


def load():
    """Load libhfmi.so (built in-tree by hippyflow_amd._build).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hippyflow_amd has no CPU fallback)" % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (res, argtypes) in NON_STATUS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = res
    _lib = lib
    # HFMI_TUNE="waves=4,nn_waves=8": the kernel tuning knobs of hfmi_tuning_set (include/hfmi.h) for a whole process -- how the A/B
    # runs of bench.py / rocprofv3 select a variant without a code change (scripts/tn_tile_ab.sh)
    for item in filter(None, os.environ.get("HFMI_TUNE", "").split(",")):
        key, _, val = item.partition("=")
        check(lib.hfmi_tuning_set(key.strip().encode(), int(val)))
    return lib


def check(status):
    if status != 0:
        raise HfmiError(status, load().hfmi_last_error().decode("utf-8", "replace"))


def call(name, *args):
    check(getattr(load(), name)(*args))


def build_tag():
    return load().hfmi_build_tag().decode()


def device_count():
    n = C.c_int(0)
    call("hfmi_device_count", C.byref(n))
    return n.value


def as_f64(a, order="C"):
    return np.require(a, dtype=np.float64, requirements=["C_CONTIGUOUS" if order == "C" else "F_CONTIGUOUS", "ALIGNED"])


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _PinnedOwner:
    def __init__(self, p):
        self.p = p

    def __del__(self):
        try:
            load().hfmi_host_free_pinned(self.p)
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """numpy array over page-locked host memory (hfmi_host_alloc_pinned): the source of ``MultiVector.upload_async``."""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    call("hfmi_host_alloc_pinned", C.c_size_t(max(nbytes, 8)), C.byref(p))
    owner = _PinnedOwner(p)
    buf = (C.c_char * max(nbytes, 8)).from_address(p.value)
    arr_owner = np.ndarray.__new__(_PinnedArray, shape, dtype, buffer=buf)
    arr_owner._owner = owner
    return arr_owner



class _PinnedArray(np.ndarray):
    """ndarray that keeps its pinned allocation alive (views share ``base``)."""
    _owner = None

    def __array_finalize__(self, obj):
        if obj is not None:
            self._owner = getattr(obj, "_owner", None)


_RANK_VARS = ("RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "PMIX_RANK", "SLURM_PROCID")
_LOCAL_RANK_VARS = ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID")


def _env_int(names, default=0):
    for name in names:
        v = os.environ.get(name)
        if v is not None and v.strip().lstrip("-").isdigit():
            return int(v)
    return default


def launcher_rank():
    """World rank of this process as its launcher exported it: torchrun ($RANK), Open MPI, MPICH / Intel MPI (PMI), PMIx, srun.
    The reference seeds hp.parRandom per MPI rank; mpirun / srun set neither $RANK nor $LOCAL_RANK."""
    return _env_int(_RANK_VARS)


def launcher_local_rank():
    """Node-local rank (picks the GPU of a one-process-per-GPU run) under the same launchers."""
    return _env_int(_LOCAL_RANK_VARS)


class Context:
    """One per GPU (hfmi_ctx).  `Context.default()` gives the process-wide context of
    cuda:<node-local rank> (one process per GPU; torchrun, mpirun and srun launches alike)."""
    _default = None

    def __init__(self, device=0):
        self.handle = C.c_void_p()
        call("hfmi_ctx_create", int(device), C.byref(self.handle))
        self.device = int(device)

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = cls(launcher_local_rank() % max(device_count(), 1))
        return cls._default

    def synchronize(self):
        call("hfmi_ctx_synchronize", self.handle)

    def set_stream(self, stream_ptr):
        call("hfmi_ctx_set_stream", self.handle, C.c_void_p(stream_ptr))

    def get_stream(self):
        """The HIP stream every kernel of this context is launched on (hipStream_t as an integer)."""
        s = C.c_void_p()
        call("hfmi_ctx_get_stream", self.handle, C.byref(s))
        return s.value or 0

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        mem = C.c_int64(0)
        call("hfmi_ctx_device_info", self.handle, name, 256, C.byref(cus), C.byref(mem))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}

    def timer_start(self):
        call("hfmi_timer_start", self.handle)

    def timer_stop(self):
        ms = C.c_double(0.0)
        call("hfmi_timer_stop", self.handle, C.byref(ms))
        return ms.value

    def bench_peaks(self):
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        call("hfmi_bench_peaks", self.handle, C.byref(a), C.byref(b), C.byref(c))
        return {"mfma_f64_tflops": a.value, "fma_f64_tflops": b.value, "hbm_copy_gbs": c.value}

    def ingest_wait(self, ticket):
        """Host wait until the upload with this ticket has left its pinned buffer."""
        call("hfmi_ingest_wait", self.handle, int(ticket))

    def ingest_fence(self):
        """Compute enqueued from here on sees every block uploaded so far (device-side wait, does not block the host)."""
        call("hfmi_ingest_fence", self.handle)

    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        call("hfmi_ctx_pci_bus_id", self.handle, buf, 64)
        return buf.value.decode()

    def bench_loaded_peak(self):
        """fp64 MFMA rate while a copy kernel streams HBM on a second stream (the power-limited regime of the solve)."""
        a, b = C.c_double(0), C.c_double(0)
        call("hfmi_bench_loaded_peak", self.handle, C.byref(a), C.byref(b))
        return {"mfma_f64_tflops_while_streaming": a.value, "hbm_copy_gbs_beside_it": b.value}

    def dense_matmul(self, A, B, ta=False, tb=False):
        """op(A) op(B) of two dense host matrices on the device (the eigensolver's general fp64 MFMA product)."""
        import numpy as np
        A, B = np.asfortranarray(A, dtype=np.float64), np.asfortranarray(B, dtype=np.float64)
        M, K = (A.shape[1], A.shape[0]) if ta else A.shape
        N = B.shape[0] if tb else B.shape[1]
        if (B.shape[1] if tb else B.shape[0]) != K:
            raise ValueError("dense_matmul: inner dimensions differ")
        Cm = np.empty((M, N), order="F")
        call("hfmi_dense_matmul", self.handle, M, N, K, int(ta), int(tb), A.ctypes.data_as(_D), B.ctypes.data_as(_D), Cm.ctypes.data_as(_D))
        return Cm

    def bench_dgemm(self, A, B, ta=False, tb=False, reps=5, want_c=True):
        """C = op(A) op(B) through the eigensolver's general fp64 MFMA product; returns (C or None, average ms of one launch)."""
        import numpy as np
        A, B = np.asfortranarray(A, dtype=np.float64), np.asfortranarray(B, dtype=np.float64)
        M, K = (A.shape[1], A.shape[0]) if ta else A.shape
        N = B.shape[0] if tb else B.shape[1]
        assert (B.shape[1] if tb else B.shape[0]) == K
        Cm = np.empty((M, N), order="F") if want_c else None
        ms = C.c_double(0.0)
        call("hfmi_bench_dgemm", self.handle, M, N, K, int(ta), int(tb), int(reps), A.ctypes.data_as(_D), B.ctypes.data_as(_D),
             Cm.ctypes.data_as(_D) if want_c else None, C.byref(ms))
        return Cm, ms.value

    def bench_hbm_read(self):
        a = C.c_double(0)
        call("hfmi_bench_hbm_read", self.handle, C.byref(a))
        return {"hbm_read_gbs": a.value}

    def bench_random_peaks(self):
        """fp64 MFMA rate on Gaussian (full-mantissa) operands rotated through the registers, alone and beside the streaming copy."""
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        call("hfmi_bench_random_peaks", self.handle, C.byref(a), C.byref(b), C.byref(c))
        return {"mfma_f64_tflops_random_operands": a.value, "mfma_f64_tflops_random_operands_while_streaming": b.value,
                "hbm_copy_gbs_beside_random_operands": c.value}

    def profile_begin(self):
        call("hfmi_profile_begin", self.handle)

    def profile_end(self):
        """One record per distinct (MFMA kernel, shape) launched since profile_begin: total ms, launches and the
        algorithmic flops / bytes of one launch."""
        G = 64
        ng = C.c_int(0)
        kind, shape = (C.c_int * G)(), (C.c_int64 * (3 * G))()
        ms, n, fl, by = (C.c_double * G)(), (C.c_int64 * G)(), (C.c_double * G)(), (C.c_double * G)()
        call("hfmi_profile_end", self.handle, G, C.byref(ng), kind, shape, ms, n, fl, by)
        names = ("k_tsgemm_tn", "k_tsgemm_nn")
        return [{"kernel": names[kind[g]], "m": shape[3 * g], "k": shape[3 * g + 1], "N": shape[3 * g + 2], "ms": ms[g],
                 "launches": int(n[g]), "flops_per_launch": fl[g], "bytes_per_launch": by[g]} for g in range(ng.value)]

    PHASES = ("apply_A", "apply_Binv", "orthogonalize", "rayleigh_quotient", "small_eig", "back_transform", "allreduce",
              "host_d2h_wait", "host_function", "host_h2d", "allreduce_overlapped")

    def profile_phases(self):
        """Milliseconds per phase of the fused solves between profile_begin and profile_end (call after profile_end).
        ``allreduce`` is inside ``apply_A`` / ``rayleigh_quotient``; the ``host_*`` legs (wall clock) are inside the
        phase whose operator is a host callback, normally ``apply_Binv``."""
        out = (C.c_double * len(self.PHASES))()
        call("hfmi_profile_phases", self.handle, out)
        return {name: out[i] for i, name in enumerate(self.PHASES)}

    def close(self):
        if self.handle:
            load().hfmi_ctx_destroy(self.handle)
            self.handle = C.c_void_p()
