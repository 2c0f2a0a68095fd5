"""Instruction count and mix of the Jacobi step pair (the loop with the 114 DPP moves) in the ISA of the production wave kernel.
Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only scale-letkf_amd/csrc/letkf_wave.hip -o /tmp/wave.s; python tools/isa_step_pair.py /tmp/wave.s"""
import sys
from collections import Counter
def steppair(path, kern="_ZN5letkf17letkf_wave_kernelILi50ELi11ELb0ELi1ELb0EEEvNS_9PointArgsE:"):
    L=open(path).read().split("\n")
    start=[i for i,l in enumerate(L) if l.startswith(kern)][0]
    end=[i for i,l in enumerate(L) if i>start and l.startswith("_ZN5letkf17letkf_wave_kernel")][0]
    K=L[start:end]
    lab={l.split(":")[0]:i for i,l in enumerate(K) if l.startswith(".LBB")}
    best=(0,None)
    for i,l in enumerate(K):
        t=l.split()
        if t and t[0].startswith("s_cbranch") and t[-1] in lab and lab[t[-1]]<i:
            body=K[lab[t[-1]]:i+1]
            nd=sum("v_mov_b32_dpp" in x for x in body)
            if nd>best[0] and nd<200: best=(nd,body)
    body=best[1]
    ins=[l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
    return len(ins), Counter(ins), body
if __name__=="__main__":
    for p in sys.argv[1:]:
        n,c,b=steppair(p)
        print(p,n,c.most_common(16))
