import sys, math, ctypes as C, torch
sys.path.insert(0, '.')
from tiny_flash_attention_amd import _lib, ops
dev = torch.device('cuda:0')
B,H,N,D,causal = 4,32,4096,128,True
mk = lambda: torch.empty((B,H,N,D), dtype=torch.float32, device=dev).normal_(0,0.5).to(torch.bfloat16)
q,k,v,do = mk(),mk(),mk(),mk()
sc = 1/math.sqrt(D)
out,lse = ops.flash_attn_fwd(q,k,v,causal,sc)
dq,dk,dv = torch.empty_like(q),torch.empty_like(k),torch.empty_like(v)
delta = torch.empty_like(lse)
p = ops.make_bwd_params(q,k,v,out,lse,do,dq,dk,dv,delta,causal,sc)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = C.c_float()
for flag in (0, 2, 1, 0, 2):
    _lib.debug_bwd_split(flag)
    best = 1e9
    for r in range(3):
        _lib.check(_lib.lib().tfa_bwd_time(C.byref(p), 3, 20, s, C.byref(ms))); best = min(best, ms.value)
    print("flag", flag, "ms", round(best, 3))
_lib.debug_bwd_split(0)
