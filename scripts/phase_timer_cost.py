"""(1) Wall time of dmx_run_iterations calls with the phase timers off / on, alternating; (2) the same call behind an idle device / a busy
one, EM restarted or continued: what is slow after an idle is the device (clocks), not the EM; (3) what brings the clocks back; (4) the ramp.
GPU box: python3 scripts/phase_timer_cost.py > gpurun_out/clock_ramp.txt"""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

import bench
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

B, S, G, dp, seed = bench.WORKLOADS['em_200k_100k_64']
problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
ctx = DeviceContext(0)
ctx.set_problem(problem.n_barcodes, problem.n_variants, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
ctx.set_betas(problem.prior_betas(add_data_prior=False))
ctx.set_addition(None)
ctx.probs_from_betas(0.01, fetch=False)
ctx.estep(np.zeros(G, dtype=np.float32), with_doublets=False, fetch_logits=False, fetch_probs=False)
ctx.set_mstep_incremental(False)
ctx.set_msteps_expected(400)
ctx.run_iterations(5, 0.01)
ctx.synchronize()
for rep in range(8):
    for on in ((False, True) if rep % 2 == 0 else (True, False)):
        ctx.set_phase_timers(on)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.run_iterations(20, 0.01)
        ctx.synchronize()
        print('timers', 'on ' if on else 'off', round((time.perf_counter() - t0) / 20 * 1e3, 4), 'ms per iteration', flush=True)

# Is the first call slower because the EM is young (data) or because the device is (clocks after idle)?
ctx.set_phase_timers(False)
pen = np.zeros(G, dtype=np.float32)


def call(title, restart, idle):
    if restart:
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        ctx.run_iterations(5, 0.01)
    ctx.synchronize()
    if idle:
        time.sleep(idle)
    ctx.reset_timings()
    t0 = time.perf_counter()
    ctx.run_iterations(20, 0.01)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    levels = ctx.guard_levels()
    print(title, round(dt / 20 * 1e3, 4), 'ms per iteration; coarse E-steps', levels['coarse_steps'], 'device-timed coarse / fine pass',
          round(levels['coarse_pass_ms'], 4), round(levels['fine_pass_ms'], 4), 'redone', ctx.guard_stats()[1], flush=True)


call('EM restarted, device busy  ', True, 0)
call('EM continued, device busy  ', False, 0)
call('EM continued, 3 s idle     ', False, 3.0)
call('EM continued, device busy  ', False, 0)
call('EM restarted, 3 s idle     ', True, 3.0)
call('EM continued, 0.1 s idle   ', False, 0.1)
call('EM continued, 0.01 s idle  ', False, 0.01)

# what brings the clocks up: explicit E-steps (bench.py clock_warmup) or EM iterations?
def after(title, prepare):
    ctx.synchronize()
    time.sleep(0.5)
    prepare()
    ctx.run_iterations(5, 0.01)
    ctx.synchronize()
    ctx.reset_timings()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.run_iterations(20, 0.01)
        ctx.synchronize()
        ts.append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
    print(title, ts, 'ms per iteration (three 20-iteration calls in a row)', flush=True)


def esteps(ms):
    def run():
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(8):
                ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
            ctx.synchronize()
    return run


after('0.5 s idle, nothing                 ', lambda: None)
after('0.5 s idle, 60 ms of E-steps        ', esteps(60))
after('0.5 s idle, 300 ms of E-steps       ', esteps(300))
after('0.5 s idle, 50 EM iterations        ', lambda: ctx.run_iterations(50, 0.01))
after('0.5 s idle, 250 EM iterations       ', lambda: ctx.run_iterations(250, 0.01))

# the ramp after an idle: E-step / M-step phase times of consecutive 4-iteration calls (each ends on the fine pass)
time.sleep(3.0)
ctx.set_phase_timers(True)
line = []
t_start = time.perf_counter()
for i in range(24):
    ctx.reset_timings()
    ctx.run_iterations(4, 0.01)
    t = ctx.timings()
    line.append(f"{(time.perf_counter() - t_start) * 1e3:.0f}ms:{t['estep']['ms'] / 4:.3f}/{t['mstep']['ms'] / 4:.3f}")
print('after 3 s idle, 4-iteration calls (end of call : E-step / M-step ms):', ' '.join(line), flush=True)
