"""World sizes 2, 3 and 4 on ONE GPU: every rank is a thread with its own DeviceContext, the exchange of the library
runs over caller-provided collectives (dmx_comm_init_host; tests/thread_plane.py).  Everything on the data path
except the RCCL calls themselves is what an N-GPU run executes: the padded variant slices, the partial sums written
straight into the exchange buffer (prow), reduce-scatter -> rounded slice -> sliced P-step -> all-gather of
genotype_prob, the re-based E-step records, the all-reduce fallback for scattered SNP groups, and the Python
sharding (distributed.py) on top.  (RCCL itself is exercised with one rank in tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

from tests import fixture_io as fio
from tests.thread_plane import ThreadWorld

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('name', ['f2_synthetic_g4.npz', 'f3_small_3.npz', 'f1_synthetic_default.npz'])
def test_sharded_entry_points_match_the_reference(name, world):
    from demuxalot_amd import distributed
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                  doublet_prior=float(fx['em0_dp']))
    prior = fx.get('em0_prior_logits')
    shared = ThreadWorld(world)

    def rank_body(plane):
        learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, barcode_prior_logits=prior, **kwargs)
        logits_df, p_df = distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=float(fx['predict0_clip']),
                                                         doublet_prior=float(fx['predict0_dp']))
        return learnt.variant_betas, probs_df, logits_df, p_df

    results = shared.run(rank_body)
    want_probs = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
    ops = {op for op, _dtype, _shape in shared.collectives}
    _cuts, _rows, contiguous = distributed.exchange_slices(fx['pack_v2snp'], world)
    assert ops == ({'reduce_scatter', 'all_gather'} if contiguous else {'all_reduce'}), ops
    for betas, probs_df, logits_df, p_df in results:
        assert list(probs_df.index) == [str(b) for b in fx['barcodes']]
        # float64 re-association over ranks can move a rounding tie of a beta by one float32 ulp
        assert np.allclose(betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0) and (betas != fx['em0_learnt_betas']).sum() <= 3
        assert np.array_equal(probs_df.values.argmax(1), want_probs.argmax(1))
        assert np.abs(probs_df.values - want_probs).max() <= 1e-5
        # predict needs no exchange: the gathered rows are the reference's, bit for bit
        fio.assert_bitwise(logits_df.values, fx['predict0_logits'], f'{name} predict logits, world {world}')
        fio.assert_bitwise(p_df.values, fx['predict0_probs'], f'{name} predict posteriors, world {world}')
    for other in results[1:]:  # every rank holds the same answers
        assert np.array_equal(other[0], results[0][0]) and np.array_equal(other[1].values, results[0][1].values)


@pytest.mark.parametrize('reduce_dtype', ['f64', 'f32'])
@pytest.mark.parametrize('world', [2, 4])
def test_sharded_em_equals_the_single_context_run(world, reduce_dtype):
    """6000 barcodes x 3000 SNPs x 24 genotypes with doublets, 4 EM iterations: the posterior rows and the addition
    of the sharded run against one context holding everything."""
    from demuxalot_amd import distributed, synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(6000, 3000, 24, calls_per_barcode=80, seed=12)
    betas = p.prior_betas()
    G = 24
    K = G * (G + 1) // 2
    pen = np.zeros(K, dtype=np.float32)
    pen[G:] = np.float32(-1.5)
    with DeviceContext(0) as ctx:
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        _l, want_probs, want_add = ctx.em(4, 0.01, pen, True, fetch_logits=False)
    shared = ThreadWorld(world)

    def rank_body(plane):
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong,
                                   reduce_dtype=reduce_dtype)
        try:
            probs, addition = em.learn(4, 0.01, pen, True)
            return em.lo, em.hi, probs, addition
        finally:
            em.ctx.close()

    results = shared.run(rank_body)
    assert [r[0] for r in results][0] == 0 and results[-1][1] == p.n_barcodes
    # float32 partial sums: each rank's rounding moves the betas by an ulp, which three more iterations amplify
    rtol, atol = (3e-7, 1e-12) if reduce_dtype == 'f64' else (1e-4, 1e-7)
    for lo, hi, probs, addition in results:
        assert np.array_equal(probs.argmax(1), want_probs[lo:hi].argmax(1))
        assert np.abs(probs - want_probs[lo:hi]).max() <= 1e-5
        assert np.allclose(addition, want_add, rtol=rtol, atol=atol), np.abs(addition - want_add).max()
        assert np.array_equal(addition, results[0][3])  # identical on every rank
    kinds = {(op, dtype) for op, dtype, _shape in shared.collectives}
    assert ('reduce_scatter', 'float64' if reduce_dtype == 'f64' else 'float32') in kinds and ('all_gather', 'float32') in kinds


def _socket_rank(rank, world, port, out):
    from demuxalot_amd import _lib, distributed
    from demuxalot_amd.plane import SocketControlPlane
    plane = SocketControlPlane(rank, world, '127.0.0.1', port=port, host_collectives=True)  # the exchange over sockets, no RCCL
    try:
        report = {}
        for name in ('f2_synthetic_g4.npz', 'f3_small_3.npz'):
            fx = fio.load(name)
            calls, genotypes, handler = fio.product_inputs(fx)
            kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                          doublet_prior=float(fx['em0_dp']))
            learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, device=0,
                                                           barcode_prior_logits=fx.get('em0_prior_logits'), **kwargs)
            want = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
            report[name] = (bool(np.allclose(learnt.variant_betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)),
                            bool(np.array_equal(probs_df.values.argmax(1), want.argmax(1))),
                            float(np.abs(probs_df.values - want).max()))
        report['hip_runtimes'] = _lib.runtime_info()['hip']
        out.put((rank, report))
    finally:
        plane.close()


def test_two_processes_one_gpu_exchange_over_the_socket_plane():
    """Two PROCESSES on the one GPU, control plane AND per-iteration exchange carried by the torch-free socket plane
    (demuxalot_amd/plane.py, through the caller-provided-collectives entry): the sharded learn_genotypes against the
    reference's outputs, and exactly one HIP runtime mapped in each worker."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_socket_rank, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    reports = [out.get(timeout=600) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for _rank, report in reports:
        assert len(report.pop('hip_runtimes')) == 1
        for name, (betas_close, argmax_same, max_dev) in report.items():
            assert betas_close and argmax_same and max_dev <= 1e-5, (name, betas_close, argmax_same, max_dev)
