import numpy as np, sys
sys.path.insert(0,'.')
from tapir_amd import engine
from oracle import oracle as orc
golden=np.load('tests/golden/reference_compute_outputs.npz')
for case in "ABC":
    r = golden[case + "_rates"]; fin = r[np.isfinite(r)]
    for k, (a, b) in enumerate(golden[case + "_intervals"]):
        integral, abserr = engine.quad_townsend(a, b, fin)
        ref=golden[case + "_site_integral"][k]; referr=golden[case + "_site_abserr"][k]
        bad=np.flatnonzero(np.abs(integral-ref) > 1e-13*np.abs(ref)+1e-28)
        for i in bad[:5]:
            o=orc.quad_townsend(a,b,fin[i])
            print(case,a,b,'rate',fin[i],'gpu',integral[i],abserr[i],'ref',ref[i],referr[i],'orc',o)
