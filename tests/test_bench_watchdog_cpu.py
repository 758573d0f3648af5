"""bench.py --gpus N without a launcher: the parent (which never touches a GPU) must survive a rank that hangs - as the first
real multi-rank RCCL run may, inside a rendezvous or a collective - by killing the rank processes it started at its deadline,
naming every rank's last phase, and exiting non-zero.  Runs without a GPU: the stalled rank stops before anything loads the
library, the other waits for it in the control plane's rendezvous."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_stalled_rank_ends_the_run_at_the_deadline():
    env = dict(os.environ, DEMUXALOT_BENCH_STALL='1:start')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    t0 = time.monotonic()
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--deadline', '6'],
                         env=env, cwd=ROOT, capture_output=True, timeout=120)
    elapsed = time.monotonic() - t0
    err = run.stderr.decode(errors='replace')
    assert run.returncode != 0, err
    assert elapsed < 60, elapsed
    assert 'deadline of 6 s passed' in err, err
    assert 'rank 1: exit code -9, last phase: start' in err, err   # the stalled one, killed by the parent
    assert 'rank 0: exit code' in err and 'control plane rendezvous' in err, err  # where the other one was waiting for it
    assert run.stdout.decode().strip() == ''  # no JSON line from a run that did not finish


def test_a_rank_that_dies_takes_the_others_down_within_the_grace_period():
    """Rank 1 fails at once (a workload name that does not exist makes every rank exit from argparse: all die, no hang), and a
    run where one rank dies while the other waits ends 30 s after the death, well before the deadline."""
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--deadline', '100', '--workload', 'em_20k_10k_64'],
                         env=dict(env, DEMUXALOT_AMD_LIB='/nonexistent/libdemux_hip.so'), cwd=ROOT, capture_output=True, timeout=200)
    err = run.stderr.decode(errors='replace')
    assert run.returncode != 0, err
    assert 'exit codes' in err and 'last phase' in err, err
