"""Host-side throughput of the exact-mode plan pipeline (emx_host_plan_mt_stream) vs the serial twin; no GPU needed.
usage: mt_pipe_bench.py [nwalkers] [nsteps] [move kind 0|1|2]"""
import sys, ctypes as C, numpy as np, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from emcee_amd import _lib
from emx_testlib import HostMT
lib=_lib.load()
N=int(sys.argv[1]) if len(sys.argv)>1 else 65536
nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 64
kind=int(sys.argv[3]) if len(sys.argv)>3 else 0
md=_lib.MoveDesc(kind,4 if kind==2 else 2,1,0,2.0,1e-5,0.2,1.7)
arr=(_lib.MoveDesc*1)(md)
for workers in (1,2,3,4,6):
    m=HostMT(np.random.RandomState(5).get_state())
    sec=C.c_double()
    rc=lib.emx_host_plan_mt_stream(m.h,N,64,1,arr,np.array([1.0]),nsteps,workers,16,None,None,None,None,None,None,None,C.byref(sec))
    print("workers",workers,"rc",rc,"ms/step %.3f"%(sec.value*1e3/nsteps))
m=HostMT(np.random.RandomState(5).get_state())
t0=time.perf_counter()
for _ in range(10): m.plan(N,64,md)
print("serial ms/step %.3f"%((time.perf_counter()-t0)*100))
