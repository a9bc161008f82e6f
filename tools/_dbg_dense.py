import os, sys, torch, json
sys.path.insert(0, "/root/repo")
os.environ["QSX_AGG_JIT_SYNC"]="1"; os.environ["QSX_AGG_JIT_MIN_ROWS"]="0"
import quickstep_amd.capi as capi
from quickstep_amd import types as T
dev=torch.device("cuda:0")
n=100_000_000
val=torch.rand(n,device=dev,dtype=torch.float64)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b)/reps,3)
for E in (25, 1000, 8000):
    keys=torch.randint(0,E,(n,),device=dev,dtype=torch.int32)
    line={"entries":E}
    for strat,name in ((T.AGG_GENERIC,"generic"),(T.AGG_COLLISION_FREE,"dense")):
        for aggs,tag in (([(T.AGG_SUM,T.col(1)),(T.AGG_COUNT_STAR,None)],"sum_count"),([(T.AGG_SUM,T.col(1))],"sum")):
            cfg=T.make_agg_config(strat,[(T.INT,None),(T.DOUBLE,None)],keys=[0],aggs=aggs,est_groups=E,num_entries=E)
            st=capi.AggState(cfg)
            line[f"{name}_{tag}_update_only_ms"]=timed(lambda: st.update([keys,val],n))
            st.close()
    print(json.dumps(line),flush=True)
