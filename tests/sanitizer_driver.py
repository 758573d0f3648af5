"""Runs INSIDE a subprocess whose LD_PRELOAD carries the AddressSanitizer runtime (tests/test_host_sanitizers.py): drives
demuxalot_amd/libdemux_host_asan.so - the host shim of the C ABI (dmx_api.cpp, pack_host.cpp) with every GPU-side symbol
stubbed at link time (csrc/host_stubs.cpp: device memory is malloc memory, the kernels' launchers do nothing) - through

  1. a fuzz of dmx_pack_calls_host against the oracle's match + de-duplication (bit-exact),
  2. a fuzz of dmx_exchange_slices against its definition,
  3. whole "runs" of the context API (problem, betas, P / E / M steps, fused EM, results, timers, cache) on random
     problems, single and with host-staged collectives of 2 .. 5 ranks (variant-sharded M-step, reduce-scatter and
     all-reduce exchange),
  4. the error contract: bad sizes, null pointers, calls out of order.

Results of (3) are meaningless (no kernel runs); what counts is that AddressSanitizer / UBSan stay silent - any report
aborts the process.  Prints 'sanitizer driver ok' at the end."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from demuxalot_amd import _lib  # noqa: E402

LIB = os.path.join(ROOT, 'demuxalot_amd', 'libdemux_host_asan.so')


def load():
    lib = ctypes.CDLL(LIB)
    for name, (res, args) in {**_lib.SIGNATURES, **_lib.DEBUG_SIGNATURES}.items():
        if hasattr(lib, name):  # the .hip translation units' entry points (results, aggregate_on_snps) are not in this build
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    _lib._lib = lib  # DeviceContext and friends now talk to the sanitizer build
    return lib


def fuzz_pack(lib, oracle, rng, rounds):
    for _ in range(rounds):
        n_chrom = int(rng.integers(1, 4))
        V = int(rng.integers(0, 60))
        var_chrom = rng.integers(0, n_chrom, size=V).astype(np.int32)
        var_pos = rng.integers(0, 25, size=V).astype(np.int32)
        var_base = rng.integers(0, 4, size=V).astype(np.uint8)
        key = var_chrom.astype(np.int64) * 1000 + var_pos * 8 + var_base
        _, first = np.unique(key, return_index=True)  # variant keys are unique (var2varid is a dict)
        first.sort()
        var_chrom, var_pos, var_base = var_chrom[first], var_pos[first], var_base[first]
        V = len(first)
        n = int(rng.integers(0, 400))
        call_chrom = rng.integers(0, n_chrom + 1, size=n).astype(np.int32)  # + a chromosome without variants
        call_pos = rng.integers(0, 27, size=n).astype(np.int32)
        call_base = rng.integers(0, 4, size=n).astype(np.uint8)
        call_cb = rng.integers(0, 9, size=n).astype(np.int32)
        call_p = rng.choice([0.0, 1e-38, 1e-4, 0.02, 0.5, 1.0], size=n).astype(np.float32)
        # the reference processes calls chromosome by chromosome (dict order): give them in that order
        order = np.argsort(call_chrom, kind='stable')
        call_chrom, call_pos, call_base, call_cb, call_p = (a[order] for a in (call_chrom, call_pos, call_base, call_cb, call_p))
        call_variant = np.empty(n, dtype=np.int32)
        out_v, out_cb = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        out_p, out_count = np.empty(n, dtype=np.float32), np.empty(n, dtype=np.int64)
        mol = np.zeros(V, dtype=np.int64)
        n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(lib.dmx_pack_calls_host(V, _lib.ptr(var_chrom), _lib.ptr(var_pos), _lib.ptr(var_base), n, _lib.ptr(call_chrom),
                                           _lib.ptr(call_pos), _lib.ptr(call_base), _lib.ptr(call_cb), _lib.ptr(call_p),
                                           _lib.ptr(call_variant), ctypes.byref(n_matched), ctypes.byref(n_unique), _lib.ptr(out_v),
                                           _lib.ptr(out_cb), _lib.ptr(out_p), _lib.ptr(out_count), _lib.ptr(mol)))
        # the oracle: exact (chromosome, position, base) lookup, then de-duplication in call order
        lookup = {(int(c), int(p), int(b)): i for i, (c, p, b) in enumerate(zip(var_chrom, var_pos, var_base))}
        want_variant = np.array([lookup.get((int(c), int(p), int(b)), -1) for c, p, b in zip(call_chrom, call_pos, call_base)], dtype=np.int32).reshape(n)
        assert np.array_equal(call_variant, want_variant)
        keep = want_variant != -1
        v, cb, p, counts = oracle.dedupe_calls(want_variant[keep], call_cb[keep], call_p[keep])
        k = n_unique.value
        assert n_matched.value == int(keep.sum()) and k == len(v)
        assert np.array_equal(out_v[:k], v) and np.array_equal(out_cb[:k], cb) and np.array_equal(out_count[:k], counts)
        assert np.array_equal(out_p[:k].view(np.uint32), p.view(np.uint32))
        assert np.array_equal(mol, np.bincount(want_variant[keep], minlength=V))


def fuzz_slices(lib, rng, rounds):
    for _ in range(rounds):
        S = int(rng.integers(0, 40))
        sizes = rng.integers(1, 5, size=S)
        v2snp = np.repeat(np.arange(S, dtype=np.int32), sizes)
        if S and rng.random() < 0.3:  # scattered SNP groups
            v2snp = rng.permutation(v2snp).astype(np.int32)
        V = len(v2snp)
        n = int(rng.integers(1, 9))
        cuts = np.zeros(n + 1, dtype=np.int64)
        rows, contiguous = ctypes.c_int64(0), ctypes.c_int32(0)
        _lib.check(lib.dmx_exchange_slices(V, _lib.ptr(v2snp), n, _lib.ptr(cuts), ctypes.byref(rows), ctypes.byref(contiguous)))
        assert cuts[0] == 0 and cuts[-1] == V and (np.diff(cuts) >= 0).all()
        assert rows.value == max(1, int(np.diff(cuts).max()))
        runs = 1 + int((np.diff(v2snp) != 0).sum()) if V else 0
        assert bool(contiguous.value) == (runs == len(np.unique(v2snp)))
        if contiguous.value:
            for c in cuts[1:-1]:
                assert c == 0 or c == V or v2snp[c] != v2snp[c - 1]  # every cut at the first variant of a SNP


def random_problem(rng, B, V, G, n):
    variant = rng.integers(0, max(V, 1), size=n).astype(np.int32)
    cb = rng.integers(0, max(B, 1), size=n).astype(np.int32)
    key = variant.astype(np.int64) * (B + 1) + cb
    _, first = np.unique(key, return_index=True)
    first.sort()
    variant, cb = variant[first], cb[first]
    p = rng.random(len(first)).astype(np.float32)
    if rng.random() < 0.5:  # hot variants: several work items
        hot = rng.integers(0, max(V, 1))
        extra = np.setdiff1d(np.arange(B, dtype=np.int32), cb[variant == hot])
        variant = np.concatenate([variant, np.full(len(extra), hot, dtype=np.int32)])
        cb = np.concatenate([cb, extra])
        p = np.concatenate([p, rng.random(len(extra)).astype(np.float32)])
    sizes = rng.integers(1, 4, size=V)
    v2snp = np.repeat(np.arange(V, dtype=np.int32), sizes)[:V]
    if rng.random() < 0.2:
        v2snp = rng.permutation(v2snp).astype(np.int32)
    return variant, cb, p, np.ascontiguousarray(v2snp, dtype=np.int32)


def run_contexts(rng, rounds):
    from demuxalot_amd._lib import DemuxHipError
    from demuxalot_amd.device import DeviceContext
    for r in range(rounds):
        B, V, G = int(rng.integers(1, 300)), int(rng.integers(1, 120)), int(rng.choice([1, 2, 5, 8, 31, 64, 65, 130]))
        variant, cb, p, v2snp = random_problem(rng, B, V, G, int(rng.integers(0, 6000)))
        doublets = bool(rng.random() < 0.4) and G > 1
        K = G * (G + 1) // 2 if doublets else G
        world = int(rng.choice([1, 1, 2, 3, 5]))
        rank = int(rng.integers(0, world))
        ctx = DeviceContext(0)
        try:
            os.environ['DEMUXALOT_AMD_EXCHANGE'] = str(rng.choice(['variant', 'variant', 'reduce_scatter', 'allreduce']))
            if world > 1 or rng.random() < 0.3:
                def collective(op, array):  # the other ranks "send zeros": nothing to add, nothing to fill
                    assert array.flags.writeable and array.size >= 0
                if rng.random() < 0.3:
                    ctx.comm_init_emulated(rank, world, 50., 10., reduce_dtype=str(rng.choice(['f64', 'f32'])))
                else:
                    ctx.comm_init_host(rank, world, collective, reduce_dtype=str(rng.choice(['f64', 'f32'])))
            ctx.set_estep_mode(str(rng.choice(['exact', 'guarded', 'fast'])))
            ctx.set_estep_dictionary(str(rng.choice(['never', 'auto', 'always'])))
            ctx.set_problem(B, V, G, variant, cb, p, v2snp)
            ctx.set_betas(rng.random((V, G)).astype(np.float32))
            ctx.set_addition(None if rng.random() < 0.5 else rng.random((V, G)).astype(np.float32))
            ctx.probs_from_betas(0.01)
            pen = np.zeros(K, dtype=np.float32)
            prior = None if rng.random() < 0.6 else rng.random((B, K)).astype(rng.choice([np.float32, np.float64]))
            ctx.estep(pen, with_doublets=doublets, prior_logits=prior)
            ctx.mstep(2. if rng.random() < 0.7 else 1.5)
            ctx.set_phase_timers(bool(rng.random() < 0.5))
            ctx.em(int(rng.integers(1, 4)), 0.01, pen, with_doublets=doublets, prior_logits=prior)
            ctx.run_iterations(2, 0.01)
            ctx.get_logits(), ctx.get_probs(), ctx.get_addition(), ctx.get_assignments()
            ctx.set_logits_needed(False)  # a call that returns no logits: they are refused afterwards, loudly, until an E-step keeps them
            ctx.run_iterations(1, 0.01)
            try:
                ctx.get_logits()
                raise AssertionError('logits served although dmx_set_logits_needed(0) was in force')
            except DemuxHipError as exc:
                assert 'logits were not kept' in str(exc), exc
            ctx.get_probs()
            ctx.set_logits_needed(True)
            ctx.run_iterations(1, 0.01)
            ctx.get_logits(), ctx.guard_probes()
            if ctx.guard_stats()[2] > 0:  # (a guarded E-step has run: there are pass times to overwrite)
                ctx.debug_set_pass_ms(coarse=0.5, fine=1.0, exact=2.0)
            ctx.get_block('probs', 0, B, 0, min(K, 3))
            ctx.set_probs(rng.random((V, G)).astype(np.float32))
            ctx.probs_from_betas_f64(rng.random((V, G)), 0.01)
            ctx.timings(), ctx.reset_timings(), ctx.guard_stats(), ctx.redo_count(), ctx.device_bytes(), ctx.estep_form()
            if r % 3 == 0:  # a second problem on the same context: the block cache hands the blocks out again
                variant2, cb2, p2, v2snp2 = random_problem(rng, B + 3, V, G, 500)
                ctx.set_problem(B + 3, V, G, variant2, cb2, p2, v2snp2)
                ctx.release_problem()
                ctx.trim_cache()
        finally:
            ctx.close()


def error_contract(lib):
    from demuxalot_amd._lib import DemuxHipError
    from demuxalot_amd.device import DeviceContext, trim_device_caches

    def refused(fn, *args, **kwargs):
        try:
            fn(*args, **kwargs)
        except (DemuxHipError, AssertionError):
            return
        raise AssertionError(f'{fn.__name__}{args[:2]} was not refused')

    ctx = DeviceContext(0)
    pen = np.zeros(4, dtype=np.float32)
    refused(ctx.estep, pen, with_doublets=False)                      # no problem yet
    refused(ctx.mstep)
    refused(ctx.run_iterations, 1, 0.01)
    refused(ctx.get_probs)
    one = np.zeros(1, dtype=np.int32)
    refused(ctx.set_problem, -1, 1, 1, one, one, np.zeros(1, dtype=np.float32), one)
    refused(ctx.set_problem, 1, 1, 0, one, one, np.zeros(1, dtype=np.float32), one)
    refused(ctx.set_problem, 1, 1, 70000, one, one, np.zeros(1, dtype=np.float32), one)     # 16-bit option encoding
    refused(ctx.set_problem, 2, 2, 2, np.array([0, 5], dtype=np.int32), np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.float32),
            np.zeros(2, dtype=np.int32))                                # variant out of range
    refused(ctx.set_problem, 2, 2, 2, np.zeros(2, dtype=np.int32), np.array([0, -1], dtype=np.int32), np.zeros(2, dtype=np.float32),
            np.zeros(2, dtype=np.int32))                                # barcode out of range
    refused(ctx.set_problem, 2, 2, 2, np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32), np.array([0.5, np.nan], dtype=np.float32),
            np.zeros(2, dtype=np.int32))                                # p_base_wrong not a probability
    refused(ctx.set_problem, 2, 2, 2, np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.float32),
            np.array([0, -3], dtype=np.int32))                          # negative SNP id
    for name, args in (('dmx_set_problem', (ctx._h, 1, 1, 1, 1, None, None, None, None)), ('dmx_create', (0, None)),
                       ('dmx_estep', (ctx._h, 0, None, None, 0, None, None)), ('dmx_device_count', (None,)),
                       ('dmx_get_block', (ctx._h, 7, 0, 0, 0, 0, None)), ('dmx_set_estep_mode', (ctx._h, 9)),
                       ('dmx_set_estep_mode', (None, 0)), ('dmx_comm_init_host', (ctx._h, 3, 2, None, None, 1)),
                       ('dmx_exchange_slices', (-1, None, 1, None, None, None)), ('dmx_trim_device_caches', (99, None)),
                       ('dmx_runtime_info', (None, 0))):
        assert getattr(lib, name)(*args) != 0, name
        assert lib.dmx_last_error()
    ctx.set_problem(3, 2, 2, np.array([0, 1, 1], dtype=np.int32), np.array([0, 1, 2], dtype=np.int32), np.array([.1, .2, .3], dtype=np.float32),
                    np.zeros(2, dtype=np.int32))
    refused(ctx.estep, np.zeros(2, dtype=np.float32), with_doublets=False)   # no genotype probabilities yet
    ctx.set_betas(np.ones((2, 2), dtype=np.float32))
    ctx.probs_from_betas(0.01)
    ctx.estep(np.zeros(2, dtype=np.float32), with_doublets=False)
    refused(ctx.get_block, 'probs', 0, 9, 0, 1)
    refused(ctx.set_betas, np.ones((3, 2), dtype=np.float32))
    ctx.close()
    ctx.close()  # twice is fine
    assert trim_device_caches(0) >= 0


def main():
    os.environ.setdefault('DEMUXALOT_AMD_ESTEP', 'exact')
    lib = load()
    from oracle import demux_oracle
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    fuzz_pack(lib, demux_oracle, rng, 4 * rounds)
    fuzz_slices(lib, rng, 10 * rounds)
    run_contexts(rng, rounds)
    error_contract(lib)
    print('sanitizer driver ok')


if __name__ == '__main__':
    main()
