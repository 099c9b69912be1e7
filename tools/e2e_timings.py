import sys, time, numpy as np, os, subprocess, tempfile
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
b = synth.sw_pairs(65536,150,150,seed=2, related_frac=0.25)
ctx.sw_score(b)
ts=[]
for _ in range(5):
    t0=time.perf_counter(); ctx.sw_score(b); ts.append(time.perf_counter()-t0)
t=min(ts); print("SW C2 one-shot agx_sw_score (plan+pack+H2D+fill+D2H): %.2f ms -> %.1f GCUPS" % (t*1e3, 65536*22500/t/1e9))
t0=time.perf_counter(); dev=ctx.sw_batch(b); t1=time.perf_counter(); dev.launch(); s=dev.scores(); t2=time.perf_counter()
print("   create %.2f ms, launch+scores %.2f ms" % ((t1-t0)*1e3,(t2-t1)*1e3))
p = synth.phmm_regions(64,64,16,100,300,seed=3)
ctx.phmm_forward(p, agx.PHMM_F32)
ts=[]
for _ in range(5):
    t0=time.perf_counter(); ctx.phmm_forward(p, agx.PHMM_F32); ts.append(time.perf_counter()-t0)
t=min(ts); print("PHMM C3 one-shot agx_phmm_forward f32: %.2f ms -> %.2f Mpairs/s" % (t*1e3, p.n_pairs/t/1e6))
with tempfile.TemporaryDirectory() as d:
    f=os.path.join(d,"sw.in"); synth.write_sw_file(f,b)
    exe="accelerating-genomics_amd/bin/antidiagonalSmithWaterman"
    t0=time.perf_counter(); o=subprocess.run([exe,f],capture_output=True); dt=time.perf_counter()-t0
    print("CLI antidiagonalSmithWaterman 65536 pairs: wall %.2f s, its own 'elapsed': %s" % (dt, o.stdout.splitlines()[-1].decode()))
    f2=os.path.join(d,"p.in"); synth.write_phmm_file(f2,p)
    exe="accelerating-genomics_amd/bin/antidiagsPairHMM"
    t0=time.perf_counter(); o=subprocess.run([exe,f2,os.path.join(d,"p.out")],capture_output=True); dt=time.perf_counter()-t0
    print("CLI antidiagsPairHMM 65536 pairs: wall %.2f s" % dt)
