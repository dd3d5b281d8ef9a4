import sys, os, ctypes as C, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import __graft_entry__ as g
pkg = g.load_package()
import torch
from bench import gen_textured_gpu
# swap in the profiling library
pkg.engine._lib = None
pkg.engine.lib_path = lambda: os.path.join(os.path.dirname(pkg.engine.__file__), os.environ.get('FCU_LIB','libfcu_prof.so'))
W,H = 3840,2160
frames = int(sys.argv[1]) if len(sys.argv)>1 else 64
nct = int(sys.argv[2]) if len(sys.argv)>2 else 2
qps=[22,27,32,37]
eng = pkg.CuEngine(W,H,max_chains=frames*4)
dev=torch.device('cuda',0)
ci=0
for f in range(frames):
    fr=gen_textured_gpu(torch,dev,W,H,seed=7+f)
    for qp in qps:
        out=torch.zeros(pkg.engine.CTU_OUT_BYTES*nct,dtype=torch.uint8,device=dev)
        eng.init_chain(ci,fr,qp=qp,out=out); ci+=1
reps=int(os.environ.get('REPS','1'))
t=time.time()
for _ in range(reps): eng.compress_chains(0,frames*4,nct//reps)
eng.sync(); dt=time.time()-t
print('chains',frames*4,'ctus',nct,'time',dt,'CTU/s',frames*4*nct/dt)
names=['rmd','pass1_total','pass1_rdoq','pass1_bits','pass2_rqt','chroma_batched','chroma_total','cu_syntax','seq_rdoq','replay','ctu_total','tu_trial_total','rqt_bits_walk','chroma_rdoq','chroma_tree_bits']
import numpy as np
acc=np.zeros(17)
packed=0
for c in range(0,frames*4,max(1,frames*4//64)):
    dc=eng.debug_counters(c)
    packed+=int(dc[15])
    acc+=np.array(dc,dtype=float)
tot=acc[10]
nsamp=len(range(0,frames*4,max(1,frames*4//64)))*nct
for i,n in enumerate(names): print('%-16s %6.2f%%  %8.2f Mticks/CTU'%(n,100*acc[i]/tot,acc[i]/nsamp/1e6))
n=len(range(0,frames*4,max(1,frames*4//64)))*nct
print('seq rdoq calls/CTU', (packed>>40)/n, 'active coefficient iterations/CTU', (packed & ((1<<40)-1))/n)
print('tu trials/CTU', acc[16]/ (len(range(0,frames*4,max(1,frames*4//64)))*nct))
