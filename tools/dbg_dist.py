import sys, os
sys.path.insert(0,'/root/repo/sh-assembly_amd'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import torch, shk, bench
dev=torch.device('cuda',0)
qb=29; K=47; L=150; R=2000000
rec=2*L+bench.NAME_W+6
offs,lens=bench.chunk_table(R,rec)
ctx=shk.Context(qb=qb,k=K,max_batch_bytes=64,max_batch_keys=R*104*2,max_batch_reads=R+1024)
genome=torch.randint(0,4,(100_000_000,),device=dev,dtype=torch.uint8)
t=bench.gen_batch_torch(torch,genome,R,L,0.00234,0,1,dev)
torch.cuda.synchronize()
dp,nw=ctx.hash_chunks(t.data_ptr(),offs,lens,on_device=True,text_bytes=t.numel())
print('nw',nw)
words=torch.as_tensor(bench._CAI(dp,nw),device=dev)
hb=qb+8
key=words & ((1<<hb)-1)
print('chunk max', int((words>>hb).max()), 'key max', hex(int(key.max())))
send=key.clone()
torch.cuda.synchronize()
try:
    st=ctx.count_words(send.data_ptr(),send.numel(),1); print(st)
except Exception as e:
    print(e, hex(ctx.last_error_bits()))
