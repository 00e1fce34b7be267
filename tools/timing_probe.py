import ctypes as C, time, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
B=10000
lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B,32,32; lay.rows=lay.cols=None; lay.mat_rows=lay.mat_cols=B*32
plan=C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
S=8
tiles=torch.rand(S*B*1024, device='cuda', dtype=torch.float64)*4.5+0.5
qv=torch.empty(S*B*1024, device='cuda', dtype=torch.float64); rv=torch.empty(S*B*528, device='cuda', dtype=torch.float64); pm=torch.empty(S*B*32, device='cuda', dtype=torch.int32)
def run(it):
    ms=C.c_float(); capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), S, it, C.byref(ms))); return ms.value
run(50); torch.cuda.synchronize()
for it in (100,400,1600):
    t0=time.perf_counter(); ms=run(it); torch.cuda.synchronize(); w=time.perf_counter()-t0
    print(it, "event ms/iter", ms, "wall ms/iter", w/it*1e3)
# direct launches timed with torch events
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0=time.perf_counter(); e0.record()
for i in range(400):
    capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), None, 0)
e1.record(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("direct: launch loop host ms", (t1-t0)*1e3, "total wall ms/iter", (t2-t0)/400*1e3, "event ms/iter", e0.elapsed_time(e1)/400)
