"""Can the default (guarded) mode be CERTIFIED across EM iterations?  

Per E-step the guard proves |posterior - reference posterior| <= 1e-5 and the same arg-max FOR THE SAME TABLE.  north_star states
the contract on the outputs of the EM loop, where the guarded run's table differs from the exact run's by what its posteriors did
to the M-steps before.  This script carries a rigorous worst-case bound through the loop, with the exact run at hand (numpy,
float64), and reports per iteration how many barcodes it can still certify - next to what the guarded run actually did.

The chain (X: exact run, Y: guarded run; delta_b: bound on |Y - X| of barcode b's posteriors, 1 when b is not certified):
  M-step   |dadd[v,g]| <= sum_{c in v} keep_c^2 (2 x_{b(c),g} delta_b + delta_b^2)            (contribution (x keep)^2, demux.py:113-118)
  P-step   |dp[v,g]|   <= (|dadd[v,g]| + p[v,g] sum_{v' in snp(v)} |dadd[v',g]|) / (den[snp,g] - sum |dadd|)   (demux.py:267-274; the clip only shrinks it)
  E-step   |dlogit[b,k]| <= sum_{c in b} keep_c |dp[v_c,k]| / (p[v_c,k] keep_c + floor_c - keep_c |dp|) =: D_table[b,k]      (demux.py:246-265)
  softmax  certified iff, with D = max_k D_table + the guard's own D_arith (0.2 taken for the coarse pass at 400 calls):
           min(x_k, 1 - x_k)(e^{2D} - 1) <= 1e-5 for every k and no second logit within 2 D of the best  -> delta_b = 1e-5, else delta_b = 1.
Four variants: `refined` (rigorous: a barcode the chain cannot certify is still computed EXACTLY on either side - by the redo here, by
the exact run there -, only on different tables, so its posteriors differ by at most max_k min(x_k, 1 - x_k)(e^{2 D_table} - 1): that
delta instead of 1, per option; what it would cost at run time: the per-option D_table is a second E-step on a table of bounds, the
per-entry |dadd| a second M-step), `optimistic` (NOT a bound: the barcodes the chain cannot certify are given delta = 1e-5 all the same - what the chain
would do if nothing poisoned it), `per option` (D_table per barcode AND option, as above - what an implementation would need one more genotype-row gather
per call for: a second E-step) and `scalar` (one eps = max |dp| per iteration times a per-barcode constant W_b = sum_c keep_c / (clip keep_c + floor_c):
free at run time).
GPU box: python3 scripts/certificate_bound.py > profiles/r6_certificate_bound.txt"""
import sys

import numpy as np

sys.path.insert(0, '.')
from demuxalot_amd import synth  # noqa: E402
from demuxalot_amd.device import DeviceContext  # noqa: E402

N_IT = 6
CLIP = 0.01


def staged(p, mode, coarse):
    ctx = DeviceContext(0)
    out = []
    try:
        ctx.set_estep_mode(mode)
        ctx.set_exact_additions(mode == 'exact')
        ctx.set_mstep_tiles('never')
        ctx.set_mstep_incremental(False)
        ctx.set_coarse_pass('always' if coarse else False)
        ctx.set_problem(p.n_barcodes, p.n_variants, p.n_genotypes, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas())
        ctx.set_addition(None)
        pen = np.zeros(p.n_genotypes, dtype=np.float32)
        for _ in range(N_IT):
            prob = ctx.probs_from_betas(CLIP)
            logits, post = ctx.estep(pen, with_doublets=False)
            out.append((prob.astype(np.float64), logits.astype(np.float64), post.astype(np.float64)))
            ctx.mstep(2.0, fetch=False)
    finally:
        ctx.close()
    return out


def analyse(name, p, d_arith):
    exact = staged(p, 'exact', False)
    guarded = staged(p, 'guarded', True)
    B, V, G = p.n_barcodes, p.n_variants, p.n_genotypes
    v, b, e = p.variant_id.astype(np.int64), p.compressed_cb.astype(np.int64), p.p_base_wrong.astype(np.float64)
    keep, floor = 1.0 - e, np.maximum(e, 1e-4)
    snp = p.v2snp.astype(np.int64)
    prior = p.prior_betas().astype(np.float64)
    W = np.bincount(b, weights=keep / (CLIP * keep + floor), minlength=B)
    print(f'## {name}: {B} barcodes x {V} variants x {G} genotypes, {len(v)} calls; W_b = sum_c keep / (clip keep + floor): median {np.median(W):.0f}, max {W.max():.0f}')
    for variant in ('refined', 'optimistic', 'per option', 'scalar'):
        delta = np.zeros(B)   # iteration 0: the same table on both sides; the guard's 1e-5 (kept) or 0 (redone exactly)
        delta[:] = 1e-5
        print(f'# bound: {variant}')
        for it in range(N_IT):
            prob_x, logit_x, x = exact[it]
            y = guarded[it][2]
            actual = np.abs(y - x).max(axis=1)
            flips = int((y.argmax(1) != x.argmax(1)).sum())
            if it > 0:
                # addition the exact run formed in the M-step before this iteration, and the bound on the guarded run's deviation from it
                xb = exact[it - 1][2]
                contrib = (keep * keep)[:, None] * (2.0 * xb[b] * delta[b][:, None] + (delta[b] ** 2)[:, None])
                dadd = np.zeros((V, G))
                np.add.at(dadd, v, contrib)
                add = np.zeros((V, G))
                np.add.at(add, v, (xb[b] * keep[:, None]) ** 2)
                tot = prior + add
                den = np.zeros((snp.max() + 1, G))
                np.add.at(den, snp, tot)
                dden = np.zeros_like(den)
                np.add.at(dden, snp, dadd)
                room = den[snp] - dden[snp]
                dp = np.where(room > 0, np.minimum((dadd + prob_x * dden[snp]) / np.maximum(room, 1e-300), 1.0), 1.0)   # (a probability moves by 1 at most)
                if variant == 'scalar':  # noqa: E721
                    D_table = (min(dp.max(), 1.0) * W)[:, None] * np.ones((1, G))
                else:
                    t = prob_x[v] * keep[:, None] + floor[:, None]
                    per_call = np.log(t / np.maximum(t - keep[:, None] * dp[v], floor[:, None]))   # the term moves between floor and 1: |dlog| <= log(t / floor)
                    D_table = np.zeros((B, G))
                    np.add.at(D_table, b, per_call)
                D = D_table.max(axis=1) + d_arith
            else:
                dp = np.zeros((V, G))
                D = np.full(B, d_arith)
            top2 = np.sort(logit_x, axis=1)[:, -2:]
            margin_ok = (top2[:, 1] - top2[:, 0]) > 2.0 * D
            worst = (np.minimum(x, 1.0 - x).max(axis=1)) * np.expm1(np.minimum(2.0 * D, 700.0))
            certified = margin_ok & (worst <= 8e-6)
            if variant == 'refined':
                d_table = D - d_arith
                same_arithmetic = np.minimum(x, 1.0 - x).max(axis=1) * np.expm1(np.minimum(2.0 * d_table, 700.0)) + (1e-7 if it > 0 else 0.0)   # + the float32 softmax on either side
                delta = np.where(certified, 1e-5, np.minimum(1.0, same_arithmetic))
            else:
                delta = np.where(certified, 1e-5, 1e-5 if variant == 'optimistic' else 1.0)
            print(f'iteration {it}: max |dp| bound {dp.max():.3g}, D (median / max over barcodes) {np.median(D):.3g} / {D.max():.3g}; certified {certified.mean() * 100:.2f} % of the barcodes; '
                  f'ACTUAL: max |guarded - exact| posterior {actual.max():.2e}, assignments that differ {flips}, barcodes beyond 1e-5: {int((actual > 1e-5).sum())}'
                  f'{"" if (actual[certified] <= 1e-5).all() else "  !! a certified barcode differs"}', flush=True)
            if certified.mean() == 0.0 and variant not in ('optimistic', 'refined'):
                print('  (nothing left to certify: every later iteration is uncertified too)')
                break


analyse('separable donors, 400 calls per barcode', synth.generate(4000, 2000, 64, calls_per_barcode=400, seed=9001), 0.2)
analyse('sibling donors, 50 calls per barcode', synth.generate(4000, 2000, 64, calls_per_barcode=50, seed=9002, sibling_pairs=True), 0.03)
