"""ctypes binding of libdemux_hip.so (C ABI: include/demux_hip.h; test / tuning surface: include/demux_hip_debug.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is visible when a
device context is requested, the calls raise.  Host-only entry points (dmx_pack_calls_host)
work without a GPU.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DEMUXALOT_AMD_LIB') or os.path.join(_HERE, 'libdemux_hip.so')

DMX_F32, DMX_F64 = 0, 1
T_PSTEP, T_ESTEP, T_MSTEP, T_MCOMBINE, T_ALLREDUCE, T_COUNT = 0, 1, 2, 3, 4, 5
TIMER_NAMES = ('pstep', 'estep', 'mstep', 'mcombine', 'allreduce')
# int (*dmx_host_collective)(void *user, int op, void *buf, int64_t count, int dtype)
HOST_COLLECTIVE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)
COLL_ALL_REDUCE, COLL_REDUCE_SCATTER, COLL_ALL_GATHER = 0, 1, 2
UNIQUE_ID_BYTES = 128

# every symbol include/demux_hip.h declares: (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    'dmx_last_error': (c_char_p, []),
    'dmx_version': (c_char_p, []),
    'dmx_device_count': (c_int, [POINTER(c_int)]),
    'dmx_create': (c_int, [c_int, POINTER(_P)]),
    'dmx_destroy': (c_int, [_P]),
    'dmx_synchronize': (c_int, [_P]),
    'dmx_pack_calls_host': (c_int, [c_int64, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P,
                                    POINTER(c_int64), POINTER(c_int64), _P, _P, _P, _P, _P]),
    'dmx_hash_host': (c_int, [_P, c_int64, c_int32, POINTER(ctypes.c_uint64)]),
    'dmx_set_problem': (c_int, [_P, c_int64, c_int64, c_int32, c_int64, _P, _P, _P, _P]),
    'dmx_pack_and_set_problem': (c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P,
                                         POINTER(c_int64), POINTER(c_int64), _P]),
    'dmx_pack_containers_and_set_problem': (c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, c_int32,
                                                    POINTER(c_int64), POINTER(c_int64), _P]),
    'dmx_stage_containers': (c_int, [_P, _P, c_int32]),
    'dmx_pack_staged_and_set_problem': (c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, c_int32,
                                                POINTER(c_int64), POINTER(c_int64), _P]),
    'dmx_get_packed_calls': (c_int, [_P, _P, _P, _P, _P]),
    'dmx_set_betas': (c_int, [_P, _P]),
    'dmx_set_prior_betas': (c_int, [_P, _P, c_double, c_int, _P, _P]),
    'dmx_set_addition': (c_int, [_P, _P]),
    'dmx_probs_from_betas': (c_int, [_P, c_float, c_float, _P]),
    'dmx_probs_from_betas_f64': (c_int, [_P, _P, c_float, c_float, _P]),
    'dmx_set_probs': (c_int, [_P, _P]),
    'dmx_estep': (c_int, [_P, c_int, _P, _P, c_int, _P, _P]),
    'dmx_mstep': (c_int, [_P, c_float, _P]),
    'dmx_em': (c_int, [_P, c_int, c_float, c_float, c_int, _P, _P, c_int, c_float, _P, _P, _P]),
    'dmx_run_iterations': (c_int, [_P, c_int, c_float, c_float, c_float]),
    'dmx_get_logits': (c_int, [_P, _P]),
    'dmx_get_probs': (c_int, [_P, _P]),
    'dmx_get_addition': (c_int, [_P, _P]),
    'dmx_get_block': (c_int, [_P, c_int, c_int64, c_int64, c_int64, c_int64, _P]),
    'dmx_get_assignments': (c_int, [_P, _P, _P]),
    'dmx_get_assignments_above': (c_int, [_P, c_float, _P, _P, POINTER(c_int64)]),
    'dmx_get_top_options': (c_int, [_P, c_int32, _P, _P]),
    'dmx_get_option_sums': (c_int, [_P, _P]),
    'dmx_set_keep_molecule_calls': (c_int, [_P, c_int]),
    'dmx_set_molecule_calls': (c_int, [_P, c_int64, _P, _P, _P]),
    'dmx_get_max_pair_count': (c_int, [_P, POINTER(c_int64)]),
    'dmx_estep_snp': (c_int, [_P, c_int, _P, c_int64, _P, c_int, _P, _P]),
    'dmx_mstep_f64': (c_int, [_P, c_double, _P]),
    'dmx_mstep_f64_sums': (c_int, [_P, c_double, _P]),
    'dmx_get_prior_betas': (c_int, [_P, _P]),
    'dmx_get_learnt_betas': (c_int, [_P, _P]),
    'dmx_exchange_slices': (c_int, [c_int64, _P, c_int32, _P, POINTER(c_int64), POINTER(c_int32)]),
    'dmx_runtime_info': (c_int, [c_char_p, c_int64]),
    'dmx_get_exchange_mode': (c_int, [_P, POINTER(c_int32)]),
    'dmx_comm_unique_id': (c_int, [_P]),
    'dmx_comm_init': (c_int, [_P, c_int, c_int, _P, c_int]),
    'dmx_comm_init_host': (c_int, [_P, c_int, c_int, _P, _P, c_int]),
    'dmx_get_timings': (c_int, [_P, POINTER(c_double), POINTER(c_int64)]),
    'dmx_reset_timings': (c_int, [_P]),
    'dmx_set_phase_timers': (c_int, [_P, c_int]),
    'dmx_set_logits_needed': (c_int, [_P, c_int]),
    'dmx_set_lean_memory': (c_int, [_P, c_int]),
    'dmx_device_bytes': (c_int, [_P, POINTER(c_int64)]),
    'dmx_trim_cache': (c_int, [_P, POINTER(c_int64)]),
    'dmx_release_problem': (c_int, [_P]),
    'dmx_trim_device_caches': (c_int, [c_int, POINTER(c_int64)]),
    'dmx_set_exact_additions': (c_int, [_P, c_int]),
    'dmx_set_estep_mode': (c_int, [_P, c_int]),
    'dmx_get_guard_stats': (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    'dmx_set_msteps_expected': (c_int, [_P, c_int64]),
}
# every symbol include/demux_hip_debug.h declares (tests, bench.py, scripts: switches, controller read-outs, self-tests)
DEBUG_SIGNATURES = {
    'dmx_comm_init_emulated': (c_int, [_P, c_int, c_int, c_double, c_double, c_int]),
    'dmx_get_exchange_compact': (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    'dmx_get_exchange_compact_table': (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    'dmx_get_guard_probes': (c_int, [_P, POINTER(c_int64), POINTER(c_int64)]),
    'dmx_debug_set_pass_ms': (c_int, [_P, c_double, c_double, c_double]),
    'dmx_get_redo_count': (c_int, [_P, POINTER(c_int64)]),
    'dmx_set_guard_adaptive': (c_int, [_P, c_int]),
    'dmx_set_coarse_pass': (c_int, [_P, c_int]),
    'dmx_set_mstep_incremental': (c_int, [_P, c_int]),
    'dmx_get_mstep_incremental': (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    'dmx_get_guard_levels': (c_int, [_P, POINTER(c_int32), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    'dmx_get_guard_direct': (c_int, [_P, POINTER(c_int32), POINTER(c_int64), POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    'dmx_set_estep_schedule': (c_int, [_P, c_int]),
    'dmx_set_estep_dictionary': (c_int, [_P, c_int]),
    'dmx_get_estep_form': (c_int, [_P, POINTER(c_int32), POINTER(c_int32)]),
    'dmx_set_estep_packing': (c_int, [_P, c_int]),
    'dmx_set_mstep_wide_addresses': (c_int, [_P, c_int]),
    'dmx_set_mstep_tiles': (c_int, [_P, c_int]),
    'dmx_get_mstep_form': (c_int, [_P, POINTER(c_int32)]),
    'dmx_get_mstep_tiles_info': (c_int, [_P, POINTER(c_int32), POINTER(c_double)]),
    'dmx_test_logf': (c_int, [_P, _P, _P, c_int64]),
    'dmx_test_logf_hot': (c_int, [_P, _P, _P, c_int64]),
    'dmx_test_expf': (c_int, [_P, _P, _P, c_int64]),
    'dmx_test_log2_hw': (c_int, [_P, _P, _P, c_int64]),
    'dmx_test_softmax': (c_int, [_P, _P, _P, c_int64, c_int64]),
}

_lib = None


class DemuxHipError(RuntimeError):
    pass


def load():
    """Loads libdemux_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DemuxHipError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C demuxalot_amd/csrc`. demuxalot_amd has no CPU fallback.')
        # multi-process GPU work on hosts whose driver only supports dmabuf IPC (RCCL across ranks fails with
        # "hipIpcGetMemHandle: invalid argument" otherwise); read by the HSA runtime when it starts, i.e. below
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status != 0:
        raise DemuxHipError(f'libdemux_hip: {load().dmx_last_error().decode()} (status {status})')


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags.c_contiguous
    return a.ctypes.data


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def runtime_info():
    """{'hip': [files], 'rccl_mapped': [files], 'rccl_loaded': file or ''}: the HIP / RCCL runtime files this process
    has mapped (include/demux_hip.h: dmx_runtime_info).  A multi-rank worker must show exactly one 'hip'."""
    buf = ctypes.create_string_buffer(8192)
    check(load().dmx_runtime_info(buf, len(buf)))
    info = {'hip': [], 'rccl_mapped': [], 'rccl_loaded': ''}
    for line in buf.value.decode().splitlines():
        key, _, value = line.partition('=')
        if key == 'rccl_loaded':
            info[key] = value
        elif key in info:
            info[key].append(value)
    return info


def device_count():
    n = c_int(0)
    status = load().dmx_device_count(ctypes.byref(n))
    return n.value if status == 0 else 0
