import os, sys, time
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'sh-assembly_amd'))
import torch, bench, shk
dev=torch.device('cuda:0')
K,L,ERR,R=47,150,0.00234,8_000_000
qb, nd, trigger = bench.sizing(K, 119157843, 16506371070, ERR)
print('sizing', qb, nd, trigger, flush=True)
rec = 2*L+bench.NAME_W+6
offs,lens = bench.chunk_table(R, rec)
ctx = shk.Context(qb=qb,k=K,trigger=trigger,num_denoise=nd,max_batch_bytes=64,max_batch_keys=R*104+4096,max_batch_reads=R+1024)
genome = torch.randint(0,4,(119_157_843,),device=dev,dtype=torch.uint8,generator=torch.Generator(device=dev).manual_seed(2))
tot_k=0
for s in range(20):
    t = bench.gen_batch_torch(torch, genome, R, L, ERR, s*R, 1000+s, dev)
    torch.cuda.synchronize()
    try:
        st = ctx.count_chunks(t.data_ptr(), offs, lens, on_device=True, text_bytes=t.numel())
    except shk.ShkError as e:
        tt = ctx.totals(); print('step', s, 'ERROR', e, 'ndistinct', tt.ndistinct, 'nelts', tt.nelts, 'free', tt.free_pointer, 'rounds_left', tt.rounds_left, flush=True); break
    tot_k += st['kmers']
    tt = ctx.totals()
    print('step', s, st['denoise_rounds'], st['removed'], 'ndistinct', tt.ndistinct, 'nelts', tt.nelts, 'rounds_left', tt.rounds_left, 'free_ptr', tt.free_pointer, flush=True)
