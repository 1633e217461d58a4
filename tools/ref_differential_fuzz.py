"""TEST INFRASTRUCTURE TOOL (build container only: needs oracle/_ref built from /root/reference).  Differential run of the emulated reference
(tests/ref_api.RefOnlineBundler: the reference's own host classes and kernels) against the oracle frame loop over random streams of the synthetic
room (random start and stride) under random settings of the bundling switches (erosion, depth / intensity filter, local verification, local dense
term, frame invalidation mode, removal period, match-count and filter thresholds).  One line per run: the stream and switches, frames tracked, the
largest number of raw matches of any image pair (above 128 the reference keeps an arrival-order-dependent subset), the largest
pose deviation, and every mismatch of the state machine / valid flags / key-frame lists (none found; log of a run with 24 further random streams,
some with depth noise: profiles/r02_ref_fuzz.txt).

usage: python tools/ref_differential_fuzz.py"""
import os
import numpy as np, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import ref_api as R
assert R.available()
from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_app_state, default_bundling_state, intrinsics_matrix
from tests.oracle_pipeline import OraclePipeline, _minf
from tests import oracle_api as _o
_orig=_o.sift_match
MAXRAW=[0]
def _wrap(*a,**k):
    r=_orig(*a,**k); MAXRAW[0]=max(MAXRAW[0],int(r[0])); return r
_o.sift_match=_wrap
W,H,S,NF=320,240,3,10
STATES={"NONE":0,"PROCESS":1,"INVALIDATE":2}
def run(start, stride, seed=None, fl=None):
    gas=default_app_state(); gbs=default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W,H
    gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.05, 5000, 2000
    gas.s_garbageCollectionEnabled=False
    gbs.s_widthSIFT, gbs.s_heightSIFT, gbs.s_maxNumImages, gbs.s_submapSize = W,H,8,S
    if fl:
        for k,v in fl.items(): setattr(gbs if hasattr(gbs,k) else gas, k, v)
    frames=[synth.scene_room(start+stride*k,W,H) for k in range(NF)]
    if seed is not None:
        rng=np.random.default_rng(seed)
        frames=[((f[0]+rng.normal(0,0.004,f[0].shape).astype(np.float32)), f[1], f[2], f[3]) for f in frames]   # depth noise 4 mm
    Kd=frames[0][3]; K=intrinsics_matrix(Kd["fx"],Kd["fy"],Kd["mx"],Kd["my"])
    op=OraclePipeline(gas,gbs,W,H,K); op._integrate=lambda *a: None
    rb=R.RefOnlineBundler(gas,gbs,W,H,K); rtm=rb.trajectory_manager()
    bad=[]; solved=False; maxdev=0.0; nvalid=0; MAXRAW[0]=0
    for i in range(NF+3):
        if i<NF:
            d,c=frames[i][0],frames[i][1]
            raw,filt=op._ingest(d,c); rb.set_frame(d,c); rb.override_filtered_depth(filt)
            rb.process_input(); op.process_input(raw,filt,c)
            ok,T,idx,lost=rb.current_integration_frame()
            if ok!=op.last_valid or lost!=op.tracking_lost: bad.append(('valid',i,ok,op.last_valid))
            if ok and op.last_valid:
                nvalid+=1
                To=op.cur_T[op.last_processed]
                dev=float(np.abs(T-To).max()); maxdev=max(maxdev,dev)
                if (not solved and dev!=0.0) or dev>5e-4: bad.append(('pose',i,dev,solved))
            # frame loop bookkeeping (no volume)
            for m in (rtm,):
                if m.active()<gas.s_maxFrameFixes: m.generate()
                for _ in range(gas.s_maxFrameFixes):
                    f,ix,TT,_=m.top_de()
                    if f: continue
                    f,ix,TT,_=m.top_in()
                    if f: m.confirm(ix); continue
                    f,ix,o_,n_=m.top_re()
                    if f: m.confirm(ix); continue
                    break
            op._reintegrate()
            if ok: rtm.add(0,T,i)
            else: rtm.add(1,_minf(),i)
            if op.last_valid: op.tm.add_frame(0,op.cur_T[op.last_processed],i)
            else: op.tm.add_frame(1,_minf(),i)
        else:
            rb.process_input(); op.process_input(); op._reintegrate()
        rb.process(); op._bundler_process()
        solved = solved or op.num_complete>0
        st=rb.state()
        mine=dict(last_processed=op.last_processed,last_valid=int(op.last_valid),local_to_solve=op.local_to_solve,last_local_solved=op.last_local_solved,past_end=op.past_end,num_complete=op.num_complete,last_valid_complete=op.last_valid_complete,tracking_lost=int(op.tracking_lost),process_state=STATES[op.state],use_solve=int(op.use_solve),total_opt_local=op.total_opt_local)
        if st!=mine: bad.append(('state',i,{k:(st[k],mine[k]) for k in st if st[k]!=mine[k]}))
        g=rb.bundler(2); ng=g.num_frames()
        if ng!=op.glob.num_images or list(g.valid(ng))!=op.glob.valid[:ng]: bad.append(('glob',i,ng,op.glob.num_images))
    return bad, maxdev, (nvalid, MAXRAW[0])
rng=np.random.default_rng(11)
for t_ in range(10):
    fl=dict(s_erodeSIFTdepth=bool(rng.random()<0.5), s_depthFilter=bool(rng.random()<0.5), s_useLocalVerify=bool(rng.random()<0.5), s_useLocalDense=bool(rng.random()<0.7),
            s_useComprehensiveFrameInvalidation=bool(rng.random()<0.5), s_numOptPerResidualRemoval=int(rng.integers(1,4)), s_minNumMatchesLocal=int(rng.integers(3,9)),
            s_minNumMatchesGlobal=int(rng.integers(3,9)), s_maxKabschResidual2=float(rng.choice([0.0001,0.0004,0.001])), s_surfAreaPcaThresh=float(rng.choice([0.01,0.032,0.08])),
            s_verifySiftErrThresh=float(rng.choice([0.03,0.075,0.15])), s_colorFilter=bool(rng.random()<0.5))
    cfg=(int(rng.integers(0,1800)), int(rng.choice([3,6,12])), None)
    t=time.time()
    try:
        bad,maxdev,nv=run(*cfg, fl=fl)
        print(cfg,{k:(round(v,4) if isinstance(v,float) else int(v)) for k,v in fl.items()},'tracked/maxraw',nv,'maxdev %.2e'%maxdev,'MISMATCH' if bad else 'ok', bad[:2], '%.0fs'%(time.time()-t), flush=True)
    except Exception as e:
        print(cfg,fl,'EXC',repr(e)[:300], flush=True)
