"""LDS bank-conflict model of the Atari resampler's byte reads (32 banks x 4 B, a wave served as two halves of 32 lanes): cycles per
read instruction for row strides and thread-to-column mappings.  The model reproduces the measured 4.25 cycles per LDS
instruction of the shipped mapping (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, 2 of them SQ_LDS_BANK_CONFLICT).  CPU only."""
import math, itertools
def axis(ssize,dsize):
    scale=1.0/(dsize/ssize); out=[]
    for d in range(dsize):
        f1=d*scale; f2=f1+scale; cell=min(scale, ssize-f1)
        s1=math.ceil(f1); s2=math.floor(f2); s2=min(s2,ssize-1); s1=min(s1,s2)
        si=[]
        if s1-f1>1e-3: si.append(s1-1)
        for s in range(s1,s2): si.append(s)
        if f2-s2>1e-3: si.append(s2)
        out.append(si)
    return out
xs=axis(160,84); ys=axis(210,84)
x0=[t[0] for t in xs]
def cost(addrs, nb, half):
    # addrs: byte addresses of 64 lanes (None inactive). returns cycles = sum over groups of max distinct dwords per bank
    groups=[addrs[:32],addrs[32:]] if half else [addrs]
    tot=0
    for g in groups:
        banks={}
        for a in g:
            if a is None: continue
            dw=a//4; banks.setdefault(dw%nb,set()).add(dw)
        tot+=max((len(v) for v in banks.values()), default=0)
    return tot
def sim(stride, nb, half, mapping='rowmajor', block=512):
    total=0; n=0
    dw4=21; rpp=block//dw4
    for w in range(block//64):
        for p in range(4):
            for j in range(3):
                for c in range(4):
                    for k in range(3):
                        addrs=[]
                        for t in range(64):
                            tid=64*w+t
                            if tid>=rpp*dw4: addrs.append(None); continue
                            if mapping=='rowmajor': qx=tid%dw4; ry=tid//dw4
                            else: ry=tid%rpp; qx=tid//rpp
                            dy=ry+rpp*p
                            if dy>=84: addrs.append(None); continue
                            row=ys[dy][j] if j<len(ys[dy]) else ys[dy][-1]
                            addrs.append(row*stride + x0[qx*4+c]+k)
                        cy=cost(addrs,nb,half)
                        total+=cy; n+= (1 if any(a is not None for a in addrs) else 0)
    return total/n
for nb in (32,64):
  for half in (False,True):
    print('banks',nb,'half' if half else 'full', {s: round(sim(s,nb,half),2) for s in (160,164,168,172,176,180,192,196)})
    print('   colmajor', {s: round(sim(s,nb,half,'col'),2) for s in (160,164,168,176)})

def sim_cols(ncol_per_thread, nb=32, half=True, stride=160, lanes_per_row=None):
    # thread owns ncol_per_thread adjacent output columns; threads laid out column-group-major within a row group
    groups_per_row = 84 // ncol_per_thread
    lpr = lanes_per_row or groups_per_row
    nthreads = 512
    rows_par = nthreads // lpr
    passes = math.ceil(84 / rows_par)
    total=0; ninstr=0
    for w in range(8):
        for p in range(passes):
            for j in range(3):
                for c in range(ncol_per_thread):
                    for k in range(3):
                        addrs=[]
                        for t in range(64):
                            tid=64*w+t
                            g=tid%lpr; rg=tid//lpr
                            dy=rg+rows_par*p
                            if g>=groups_per_row or rg>=rows_par or dy>=84: addrs.append(None); continue
                            row=ys[dy][j]
                            addrs.append(row*stride + x0[g*ncol_per_thread+c]+k)
                        if any(a is not None for a in addrs):
                            total+=cost(addrs,nb,half); ninstr+=1
    return total, ninstr, round(total/ninstr,2)
print('current (4 cols/thread, 21 per row):', sim_cols(4))
print('2 cols/thread, 42 per row:', sim_cols(2))
print('2 cols/thread, 64 lanes per row:', sim_cols(2, lanes_per_row=64))
print('1 col/thread, 84 per row:', sim_cols(1))
print('1 col/thread, 128 lanes per row:', sim_cols(1, lanes_per_row=128))
for st in (160,164,168,176,192):
    print('stride',st,'2 cols:', sim_cols(2,stride=st), '1 col:', sim_cols(1,stride=st))
