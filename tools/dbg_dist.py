import sys, os
sys.path.insert(0,'/root/repo/sh-assembly_amd'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import torch, shk, bench
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dev=torch.device('cuda',0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
qb=29; K=47; L=150; R=2000000
rec=2*L+bench.NAME_W+6
offs,lens=bench.chunk_table(R,rec)
ctx=shk.Context(qb=qb,k=K,max_batch_bytes=64,max_batch_keys=R*104*2,max_batch_reads=R+1024)
genome=torch.randint(0,4,(100_000_000,),device=dev,dtype=torch.uint8)
hb=qb+8
t=bench.gen_batch_torch(torch,genome,R,L,0.00234,0,1,dev)
torch.cuda.synchronize()
dp,nw=ctx.hash_chunks(t.data_ptr(),offs,lens,on_device=True,text_bytes=t.numel())
raw=torch.as_tensor(bench._CAI(dp,nw),device=dev)
print('raw ptr',hex(raw.data_ptr()),hex(dp),'chunkmax raw',int((raw>>hb).max()), 'n',raw.numel())
words = (raw & ((1 << hb) - 1)) | (((raw >> hb) * 1 + 0) << hb)
print('words==raw', bool((words==raw).all()))
key = words & ((1 << hb) - 1)
owner = key >> hb
print('owner max',int(owner.max()))
order = torch.argsort(owner)
print('order ok', int(order.max()), int(order.min()), order.dtype)
send = words[order].contiguous()
print('send multiset ok', bool((send.sort().values==words.sort().values).all()), 'chunkmax', int((send>>hb).max()))
sc=torch.bincount(owner,minlength=1); print('sc',sc.tolist())
recv=torch.empty((nw,),dtype=torch.int64,device=dev)
dist.all_to_all_single(recv, send, output_split_sizes=[nw], input_split_sizes=[nw])
torch.cuda.synchronize()
print('recv==send', bool((recv==send).all()), 'first mismatch', int((recv!=send).nonzero()[0]) if not bool((recv==send).all()) else -1)
from shk import dist as shkdist
r2=shkdist.route_words(words,hb,qb,1,dev); torch.cuda.synchronize(); print('route_words ok', bool((r2.sort().values==words.sort().values).all()))
st=shkdist.ShardState(310381209,8,dev)
ctx.stage_words(r2.data_ptr(),r2.numel()); print(shkdist.sharded_count(ctx,st,len(offs)), st.ndistinct)
dist.destroy_process_group()
