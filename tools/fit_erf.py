import numpy as np
from scipy.special import erf, erfc
from numpy.polynomial import chebyshev as Ch
T=3.95
def target(t): return -np.log2(erfc(t))/t
# f(0) limit = 2/sqrt(pi)*log2(e)
def fit(deg, iters=30):
    # weighted least squares on chebyshev nodes with iterative reweighting toward minimax of erf abs error
    n=4000
    t=(np.cos(np.pi*(np.arange(n)+0.5)/n)*0.5+0.5)*T
    t=np.maximum(t,1e-6)
    f=target(t)
    # derivative of erf-reconstruction wrt f: d/df [1-2^(-t f)] = ln2 * t * 2^(-t f)
    sens=np.log(2)*t*erfc(t)
    w=sens.copy()
    for it in range(iters):
        V=np.vander(t/T*2-1,deg+1,increasing=True)
        # use chebyshev basis for conditioning
        V=Ch.chebvander(t/T*2-1,deg)
        c,*_=np.linalg.lstsq(V*w[:,None],f*w,rcond=None)
        err=(V@c-f)*sens
        w=w*(1+ 2*np.abs(err)/np.abs(err).max())
        w/=w.max()
    return c,np.abs(err).max()
for deg in range(6,13):
    c,e=fit(deg)
    print(deg,e)

print("----")
from numpy.polynomial import polynomial as P
deg=8
c,e=fit(deg,60)
# chebyshev in u = 2t/T - 1  -> monomial in t
pu=Ch.cheb2poly(c)           # poly in u
# substitute u = a t + b
a,b=2/T,-1.0
pt=np.zeros(1)
for k,ck in enumerate(pu):
    pt=P.polyadd(pt, ck*P.polypow([b,a],k))
print("monomial coeffs (t^0..):",[float(np.float32(x)) for x in pt])
co=np.array(pt,dtype=np.float32)
def erf32(x):
    x=np.asarray(x,dtype=np.float32)
    t=np.minimum(np.abs(x),np.float32(T))
    p=np.full_like(t,co[-1])
    for k in range(deg-1,-1,-1):
        p=(p.astype(np.float64)*t.astype(np.float64)+co[k].astype(np.float64)).astype(np.float32)  # fma
    y=(-(p*t)).astype(np.float32)
    E=np.exp2(y.astype(np.float64)).astype(np.float32)
    r=(np.float32(1)-E).astype(np.float32)
    return np.copysign(r,x)
x=np.linspace(-6,6,4000001).astype(np.float32)
ref=erf(x.astype(np.float64))
got=erf32(x).astype(np.float64)
ulp=np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
err=np.abs(got-ref)
print("max abs err",err.max(),"at",x[err.argmax()])
# abs err in units of 2^-24 (ulp of values in [0.5,1))
print("max err / 5.96e-8:",(err/5.96e-8).max())
# gelu comparison: ours vs reference-style float32 with correctly rounded erf
def gelu_with(e32,x):
    x=x.astype(np.float32)
    return (np.float32(0.5)*x*(np.float32(1)+e32)).astype(np.float32)
xs=(x*np.float32(np.sqrt(2))).astype(np.float32)
arg=(xs*np.float32(0.70710678118654752)).astype(np.float32)
g_ref=gelu_with(erf(arg.astype(np.float64)).astype(np.float32),xs)
g_our=gelu_with(erf32(arg),xs)
for dt,name in ((np.float16,"fp16"),):
    a=g_ref.astype(dt); b=g_our.astype(dt)
    print(name,"mismatch frac",np.mean(a!=b))
import torch
a=torch.from_numpy(g_ref).bfloat16(); b=torch.from_numpy(g_our).bfloat16()
print("bf16 mismatch frac",(a!=b).float().mean().item())
print("fp32 gelu max abs diff",np.abs(g_ref-g_our).max(), "max rel (|x|<4)", np.max(np.abs(g_ref-g_our)[np.abs(xs)<4]/np.maximum(np.abs(g_ref[np.abs(xs)<4]),1e-30)))
print("---- vs torch CPU gelu, x~N(0,s)")
import torch.nn.functional as F
for s in (1.0,2.0):
    for dt in (torch.float16, torch.bfloat16):
        xx=(torch.randn(4_000_000)*s).to(dt)          # storage dtype values
        ref=F.gelu(xx)                                 # oracle path: low-precision tensor gelu on CPU
        xf=xx.float().numpy()
        arg=(xf*np.float32(0.70710678118654752)).astype(np.float32)
        ours=torch.from_numpy((np.float32(0.5)*xf*(np.float32(1)+erf32(arg))).astype(np.float32)).to(dt)
        exact=torch.from_numpy((0.5*xf.astype(np.float64)*(1+erf(xf.astype(np.float64)/np.sqrt(2))))).to(dt)
        print(s,dt,"ours!=torch",(ours!=ref).float().mean().item(),"torch!=exact",(ref!=exact).float().mean().item(),"ours!=exact",(ours!=exact).float().mean().item())
