import numpy as np, sys, time
sys.path.insert(0,'.')
from tests._cases import *
from clonealign_amd.engine import HipEngine
from oracle.fused_numpy import FusedModel
for kw in [dict(N=300,G=130,C=3,K=1), dict(N=257,G=70,C=4,K=0), dict(N=200,G=90,C=4,K=2,P=1,S=2,extra=True), dict(N=150,G=64,C=11,K=1), dict(N=3000,G=1500,C=6,K=1)]:
    case = make_case(seed=3, **kw)
    eng = HipEngine(**case); ora = FusedModel(**case, dtype='float32')
    print(kw, eng.info())
    st = perturbed_state({n: getattr(ora,n).shape for n in ora.VAR_NAMES})
    for n,v in st.items():
        setattr(ora,n,v.astype(ora.pdt)); eng.set(n,v)
    eps = eps_for(ora.S, ora.G, 7)
    print(' terms', eng.elbo_terms(eps), ora.elbo_terms(eps))
    ge,ee = eng.gradients(eps); go,eo = ora.gradients(eps)
    print(' elbo', ee, eo)
    for n in ora.VAR_NAMES:
        if go[n].size: print('  ', n, np.abs(ge[n]-go[n]).max(), np.abs(go[n]).max())
    eng.gamma_init(eps); ora.gamma_init(eps)
    print(' ginit', np.abs(eng.get('gamma_logits')-ora.gamma_logits).max())
    for i in range(5):
        e = eps_for(ora.S, ora.G, 100+i); eng.step(e); ora.step(e)
    so,se = ora.get_state(), eng.get_state()
    for n in ora.VAR_NAMES:
        if so[n].size: print('  after5', n, np.abs(se[n]-so[n]).max(), np.abs(so[n]).max())
    eng.close()
