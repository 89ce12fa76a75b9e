"""Experiment: where does one lockstep simulation spend its time?"""
import time, sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import engine, network, ops, _lib
import ctypes as C

g=json.load(open(os.path.join(os.path.dirname(__file__),'..','tests','golden','simulate.json')))
B=1024
torch.manual_seed(0)
policy=network.SLPolicy().cuda().eval(); value=network.Value().cuda().eval()
m=engine.BatchedMCTS(B,policy,value,ops.RolloutWeights(g['shipped_w'],g['shipped_b']),capacity=8192)
own=torch.full((B,),engine.START_OWN,dtype=torch.int64,device='cuda')
opp=torch.full((B,),engine.START_OPP,dtype=torch.int64,device='cuda')
act=torch.ones(B,dtype=torch.uint8,device='cuda')
T={}
def tick(name,t0):
    torch.cuda.synchronize(); T[name]=T.get(name,0)+time.perf_counter()-t0; return time.perf_counter()
for it in range(120):
    if it==20: T.clear()
    t=time.perf_counter()
    m._select(own,opp,act,True); t=tick('select',t)
    idx=torch.nonzero(m.needs_expand & act).reshape(-1); t=tick('nonzero',t)
    if idx.numel()>0:
        games=idx.to(torch.int32)
        sp=ops.encode_planes(m.cur_own[idx],m.cur_opp[idx]); t=tick('gather+encode',t)
        with torch.no_grad(): probs=m.policy_fn(sp).contiguous()
        t=tick('policy(n=%s)'%('var'),t)
        _lib.check(_lib.lib().iago_mcts_expand(m.tree.ref(),C.c_void_p(games.data_ptr()),games.numel(),C.c_void_p(m.cur_node.data_ptr()),C.c_void_p(m.legal.data_ptr()),C.c_void_p(probs.data_ptr()),None,None)); t=tick('expand',t)
        sub=torch.zeros_like(act); sub[idx]=1
        m._select(own,opp,sub,False); t=tick('select2',t)
    ops.encode_planes(m.cur_own,m.cur_opp,out=m.planes); t=tick('encode',t)
    with torch.no_grad(): m.v=m.value_fn(m.planes).contiguous()
    t=tick('value',t)
    ops.rollout(m.cur_own,m.cur_opp,m.rollout_weights,seed=0,id_base=0,stream_id=it,out=m._rollout_out); t=tick('rollout',t)
    _lib.check(_lib.lib().iago_leaf_values(C.c_void_p(m.v.data_ptr()),C.c_void_p(m.z.data_ptr()),0.5,C.c_void_p(m.leaf_value.data_ptr()),B,None))
    _lib.check(_lib.lib().iago_mcts_backup(m.tree.ref(),C.c_void_p(act.data_ptr()),C.c_void_p(m.cur_node.data_ptr()),C.c_void_p(m.leaf_value.data_ptr()),None)); t=tick('leaf+backup',t)
tot=sum(T.values())
for k,v in sorted(T.items(),key=lambda kv:-kv[1]): print('%-18s %8.3f ms/sim'%(k,v/100*1e3))
print('total %.3f ms/sim (with per-phase syncs)'%(tot/100*1e3))
