"""Large random binary trees (run on a GPU box): the lane kernel with P through the
scalar cache (tree too large for LDS), the MFMA kernels with hundreds of steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, networkx as nx
from raoteh_amd import synth, device, _mjp_dense
from oracle import oracle_numpy as orc
ctx=device.get_context(0)
for n, nleaves, nsites in ((4, 1500, 300), (20, 700, 200), (61, 300, 64)):
    T, root, leaves = synth.random_tree(2*nleaves-1, seed=n, max_children=2)
    leaves=[v for v in T if T.degree(v)==1 and v!=root]
    rng=np.random.RandomState(n)
    for na,nb in nx.bfs_edges(T,root):
        M=rng.exponential(size=(n,n))*1e-3/n+np.eye(n); T[na][nb]['P']=M/M.sum(axis=1,keepdims=True)
    w=rng.uniform(0.1,1,size=n)
    # (no rescaling, as in the reference: random leaf states underflow on trees this
    # large; near-identity matrices and nearly constant columns stay representable)
    states=np.repeat(rng.randint(0,n,size=(nsites,1)),len(leaves),axis=1).astype(np.uint8)
    flip=rng.uniform(size=states.shape)<0.002
    states[flip]=rng.randint(0,n,size=int(flip.sum()))
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    model=device.TreeModel(T,root,n,ctx=ctx); model.set_transitions(esd); model.set_root_distn(w)
    t0=time.time()
    ll,st=model.log_likelihoods(model.upload_sites(leaves, states, kind='state'))
    t1=time.time()
    dense=np.zeros((16,len(leaves),n)); ii,kk=np.indices((16,len(leaves))); dense[ii,kk,states[:16]]=1.0
    want,wst=orc.batch_log_likelihoods(idx,ptr,esd,[pre.index(v) for v in leaves],dense,w)
    ok=wst==0
    err=np.max(np.abs(ll[:16][ok]-want[ok])/np.abs(want[ok])) if ok.any() else 0
    print(n, 'leaves', len(leaves), 'nodes', T.number_of_nodes(), 'depth', model.schedule_depth if hasattr(model,'schedule_depth') else '?', ctx.kernel_time(1)[2], 'gpu %.3fs'%(t1-t0), 'zero', int((st&1).sum()), 'err %.2e'%err, 'finite', bool(np.isfinite(want[ok]).all()))
