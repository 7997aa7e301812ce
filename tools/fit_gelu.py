#!/usr/bin/env python3
"""tools/fit_gelu.py -- the transcendental-free GELU forms measured (and rejected) in round 4: S(x) = 0.5 + t P(z),
t = clamp(x, -c, c), z = 2 t^2 / c^2 - 1, P fitted in the Chebyshev basis of z by iteratively re-weighted least squares,
evaluated in float32 Horner form.  Prints the coefficients and the largest |x S(x) - gelu(x)| on [-10, 10].
On MI355X the forms were slower than x * rcp(1 + exp2(.)): see the comment in csrc/gemm_common.h."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import ndtr
def S_tanh(x):
    u = np.sqrt(2/np.pi)*(x+0.044715*x**3)
    return 1/(1+np.exp(-2*u))
def S_erf(x): return ndtr(x)
def fit(S, c, deg, iters=200):
    n = 6000
    y = (np.cos(np.pi*(np.arange(n)+0.5)/n)*0.5+0.5)*c*c
    x = np.sqrt(y)
    F = (S(np.maximum(x,1e-7))-0.5)/np.maximum(x,1e-7)
    w = x+1e-3
    z = 2*y/(c*c)-1
    V = C.chebvander(z, deg)
    wt = np.ones(n)
    best=None
    for it in range(iters):
        coef,*_ = np.linalg.lstsq(V*(w*wt)[:,None],F*w*wt,rcond=None)
        err = (V@coef-F)*w
        m = np.abs(err).max()
        if best is None or m<best[1]: best=(coef.copy(),m)
        wt *= (1+0.5*np.abs(err)/m); wt/=wt.mean()
    return best
for name,S,c,deg in (("tanh",S_tanh,4.5,10),("erf",S_erf,5.0,11)):
    coef,e = fit(S,c,deg)
    pz = C.cheb2poly(coef).astype(np.float32)
    print(name,"c",c,"deg",deg,"fit err",e)
    print("  {"+", ".join("%.9ef"%v for v in pz)+"}")
    xs = np.linspace(-10,10,4000001).astype(np.float32)
    t = np.clip(xs,np.float32(-c),np.float32(c)); 
    z = (t*t).astype(np.float64)*np.float64(np.float32(2/(c*c)))-1.0; z=z.astype(np.float32)
    acc = np.full_like(xs,pz[-1])
    for k in range(deg-1,-1,-1):
        acc = (acc.astype(np.float64)*z.astype(np.float64)+np.float64(pz[k])).astype(np.float32)
    s = (t.astype(np.float64)*acc.astype(np.float64)+0.5).astype(np.float32)
    g = (xs*s).astype(np.float64)
    ref = xs.astype(np.float64)*S(xs.astype(np.float64))
    err = np.abs(g-ref)
    print("  max abs err %.3e at x=%.3f ; s range [%.3e, %.8f]; max |err| for x<-c: %.2e" % (err.max(), xs[err.argmax()], s.min(), s.max(), err[xs<-c].max()))
