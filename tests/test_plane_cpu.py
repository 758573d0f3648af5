"""The torch-free control plane (demuxalot_amd/plane.py) and the sharded entry points over it, on CPU: processes as
ranks, sockets as the plane, tests/cpu_context.OracleContext (oracle arithmetic; the library's exchange sequence
repeated over the plane's host collectives) in place of the device context.  No torch in any worker."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(target, world, *args, timeout=600):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, out) + args) for r in range(world)]
    for pr in procs:
        pr.start()
    results = sorted((out.get(timeout=timeout) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return [r for _rank, r in results]


def _primitives(rank, world, port, out, use_file):
    sys.path.insert(0, ROOT)
    from demuxalot_amd.plane import SocketControlPlane
    if use_file:  # the rendezvous a launcher gives: no free port known in advance, rank 0 publishes one
        os.environ['MASTER_PORT'], os.environ['TORCHELASTIC_RUN_ID'] = str(port), f'test{port}'
        plane = SocketControlPlane(rank, world, '127.0.0.1', host_collectives=True)
    else:
        plane = SocketControlPlane(rank, world, '127.0.0.1', port=port, host_collectives=True)
    report = dict(
        bcast=plane.broadcast_bytes(b'unique-id' if rank == 0 else None),
        total=plane.sum_int64(np.arange(5) * (rank + 1)).tolist(),
        slowest=plane.max_float64(1.5 * rank),
        everywhere=plane.gather_rows(np.full((rank + 1, 3), rank, dtype=np.float32)).tolist(),
        ok=plane.all_ok(rank != 1, 'boom'))
    at_root = plane.gather_to_root(np.full((2, 2), rank, dtype=np.int32))
    report['at_root'] = None if at_root is None else at_root.tolist()
    buf = np.arange(world * 4, dtype=np.float64).reshape(world, 4) * (rank + 1)
    plane.host_collective('reduce_scatter', buf)
    report['reduce_scatter'] = buf[rank].tolist()
    buf = np.zeros((world, 2), dtype=np.float32)
    buf[rank] = rank + 10
    plane.host_collective('all_gather', buf)
    report['all_gather'] = buf.tolist()
    buf = np.ones(7) * (rank + 1)
    plane.host_collective('all_reduce', buf)
    report['all_reduce'] = buf.tolist()
    plane.barrier()
    report['torch_loaded'] = 'torch' in sys.modules
    out.put((rank, report))
    plane.close()


@pytest.mark.parametrize('use_file', [False, True])
def test_socket_plane_primitives(use_file):
    world = 3
    reports = _run(_primitives, world, use_file)
    tri = sum(range(1, world + 1))
    for rank, r in enumerate(reports):
        assert r['bcast'] == b'unique-id' and r['total'] == (np.arange(5) * tri).tolist() and r['slowest'] == 3.0
        assert r['everywhere'] == [[0.] * 3] + [[1.] * 3] * 2 + [[2.] * 3] * 3
        assert r['ok'] == (False, 'rank 1: boom')
        assert r['at_root'] == ([[0, 0]] * 2 + [[1, 1]] * 2 + [[2, 2]] * 2 if rank == 0 else None)
        assert r['reduce_scatter'] == (np.arange(world * 4).reshape(world, 4)[rank] * tri).tolist()
        assert r['all_gather'] == [[10., 10.], [11., 11.], [12., 12.]] and r['all_reduce'] == [float(tri)] * 7
        assert not r['torch_loaded']


def _entry_points(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from demuxalot_amd import distributed
    from demuxalot_amd.plane import SocketControlPlane
    from tests import fixture_io as fio
    from tests.cpu_context import OracleContext
    plane = SocketControlPlane(rank, world, '127.0.0.1', port=port, host_collectives=True)
    report = {}
    for name in ('f2_synthetic_g4.npz', 'f3_small_3.npz'):  # F2: sliced exchange; F3: scattered SNP groups -> all-reduce
        fx = fio.load(name)
        calls, genotypes, handler = fio.product_inputs(fx)
        kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']), doublet_prior=float(fx['em0_dp']))
        prior = fx.get('em0_prior_logits')
        want = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
        for results in ('all', 'root'):
            learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, context_factory=OracleContext,
                                                           barcode_prior_logits=prior, results=results, **kwargs)
            here = results == 'all' or rank == 0
            assert (probs_df is not None) == here
            report[f'{name} learn {results}'] = dict(
                betas_close=bool(np.allclose(learnt.variant_betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)),
                max_dev=float(np.abs(probs_df.values - want).max()) if here else 0.,
                argmax_same=bool(np.array_equal(probs_df.values.argmax(1), want.argmax(1))) if here else True)
        # the generator, iteration by iteration (every iteration runs the exchange); the yielded addition is the one
        # the iteration's E-step used
        stages = list(distributed.staged_genotype_learning(calls, genotypes, handler, plane, context_factory=OracleContext,
                                                           barcode_prior_logits=prior, **kwargs))
        assert len(stages) == kwargs['n_iterations']
        worst, add_ok = 0., True
        for it, (frame, dbg) in enumerate(stages):
            worst = max(worst, float(np.abs(frame.values - fx[f'em0_it{it}_probs']).max()))
            add_ok = add_ok and np.allclose(dbg['genotype_addition'], fx[f'em0_it{it}_addition'], rtol=3e-7, atol=1e-12)
            assert np.array_equal(dbg['genotype_prior'], fx['pack1_betas'])
        report[f'{name} staged'] = dict(max_dev=worst, additions_close=bool(add_ok))
        logits_df, p_df = distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=float(fx['predict0_clip']),
                                                         doublet_prior=float(fx['predict0_dp']), context_factory=OracleContext)
        report[f'{name} predict'] = bool(np.array_equal(logits_df.values.view(np.uint32), fx['predict0_logits'].view(np.uint32)) and
                                         np.array_equal(p_df.values.view(np.uint32), fx['predict0_probs'].view(np.uint32)))
    report['torch_loaded'] = 'torch' in sys.modules
    out.put((rank, report))
    plane.close()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_entry_points_over_the_socket_plane(world):
    for report in _run(_entry_points, world):
        assert not report.pop('torch_loaded')
        for what, r in report.items():
            if what.endswith('predict'):
                assert r, what
            elif what.endswith('staged'):
                assert r['max_dev'] <= 1e-5 and r['additions_close'], (what, r)
            else:
                assert r['betas_close'] and r['argmax_same'] and r['max_dev'] <= 1e-5, (what, r)


def _one_rank_fails(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from demuxalot_amd import distributed
    from demuxalot_amd.plane import SocketControlPlane
    from tests import fixture_io as fio
    from tests.cpu_context import OracleContext

    class Failing(OracleContext):
        def pack_containers_and_set_problem(self, *args, **kwargs):
            if rank == 1:
                raise AssertionError('calls on a chromosome without variants')  # what only one shard may see
            return super().pack_containers_and_set_problem(*args, **kwargs)
    plane = SocketControlPlane(rank, world, '127.0.0.1', port=port, timeout=60., host_collectives=True)
    fx = fio.load('f2_synthetic_g4.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    try:
        distributed.learn_genotypes(calls, genotypes, handler, plane, context_factory=Failing, n_iterations=2)
        outcome = 'returned'
    except AssertionError as exc:
        outcome = f'AssertionError: {exc}'
    except RuntimeError as exc:
        outcome = f'RuntimeError: {exc}'
    out.put((rank, outcome))
    plane.close()


def test_a_failing_rank_fails_every_rank_instead_of_hanging_the_others():
    outcomes = _run(_one_rank_fails, 3, timeout=120)
    assert outcomes[1] == 'AssertionError: calls on a chromosome without variants'
    for other in (outcomes[0], outcomes[2]):
        assert other.startswith('RuntimeError: a rank of the sharded run failed: rank 1: AssertionError'), other
