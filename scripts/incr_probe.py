import sys, numpy as np
sys.path.insert(0,'/root/repo')
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
for (B,S,G,cpb) in ((200000,100000,64,400),):
    p = synth.generate(B,S,G,calls_per_barcode=cpb,seed=1700+G+B)
    ctx = DeviceContext(0); ctx.set_estep_mode('exact'); ctx.set_exact_additions(False); ctx.set_mstep_tiles(True)
    ctx.set_problem(p.n_barcodes,p.n_variants,G,p.variant_id,p.compressed_cb,p.p_base_wrong,p.v2snp); ctx.set_betas(p.prior_betas()); ctx.set_addition(None)
    pen=np.zeros(G,dtype=np.float32); ctx.set_phase_timers(True); ctx.reset_timings(); out=[]
    for it in range(8):
        ctx.probs_from_betas(0.01,fetch=False); ctx.estep(pen,with_doublets=False,fetch_logits=False,fetch_probs=False); ctx.mstep(2.,fetch=False)
        out.append(ctx.mstep_incremental())
    t=ctx.timings(); print(B,S,G,cpb,out, 'mstep ms', t['mstep']['ms']/t['mstep']['launches']); ctx.close()
