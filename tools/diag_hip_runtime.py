import sys, os, ctypes
def maps(tag):
    libs=set()
    for l in open('/proc/self/maps'):
        p=l.split()[-1]
        if any(k in p for k in ('amdhip','hsa-runtime','sspgpu')): libs.add(p)
    print(tag, sorted(libs), flush=True)
order=sys.argv[1]
sys.path.insert(0,'/root/repo')
if order=='torch_first':
    import torch; maps('after import torch')
    print('avail', torch.cuda.is_available(), flush=True); maps('after is_available')
    x=torch.zeros(4,device='cuda'); print('torch ok', x.sum().item(), flush=True)
    from speech_signal_processing_amd import api
    try:
        c=api.Context.for_torch(0); print('ctx ok')
    except Exception as e: print('ctx fail', e)
    maps('end')
else:
    from speech_signal_processing_amd import api
    c=api.Context(0); print('ctx ok'); maps('after ctx')
    import torch; maps('after torch')
    try:
        x=torch.zeros(4,device='cuda'); print('torch ok', x.sum().item())
    except Exception as e: print('torch fail', e)
    maps('end')
