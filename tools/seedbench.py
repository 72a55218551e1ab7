import sys, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'era-zkevm_circuits_amd'))
import numpy as np, torch, zkgl
import bench
zkgl.init(0)
cs, limit = bench.build_vm_cs(zkgl, 20)
n_outer, n_loop = cs.input_words()
st = cs.stats(); print({k:st[k] for k in ('loop_ops','seed_ops','seed_words','seed_slots')})
for B in (16, 145):
    rng=np.random.default_rng(0xC2); outer, loop = bench.vm_inputs(rng, n_outer, n_loop, B, limit)
    cs.set_batch(B)
    dev=torch.device('cuda',0)
    d_outer = torch.from_numpy(outer.view(np.int64)).to(dev); d_loop = torch.from_numpy(loop.view(np.int64)).to(dev)
    cs.bind_inputs(False, d_outer, n_outer); cs.bind_inputs(True, d_loop, n_loop)
    stream = torch.cuda.current_stream().cuda_stream
    for mode in ('0',) + (('1',) if B == 16 else ()):
        os.environ['ZKGL_SEED_GENERIC']=mode
        d_loop.copy_(torch.from_numpy(loop.view(np.int64)))
        torch.cuda.synchronize(); t=time.perf_counter(); cs.seed_carried_inputs(d_loop, stream); torch.cuda.synchronize(); dt=time.perf_counter()-t
        res = d_loop.cpu().numpy().copy()
        print('B',B,'generic' if mode=='1' else 'cone', 'seed s', round(dt,3))
        if mode=='0': cone=res
        else: print('cone == generic:', np.array_equal(cone,res))
    ok,f = cs.resolve_and_check(stream); print('ok',ok)
