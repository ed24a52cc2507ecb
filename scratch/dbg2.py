import numpy as np, sys
sys.path.insert(0,'.')
from tapir_amd import synth, engine
from oracle import oracle as orc
d=synth.simulate(10,333,5,13)
pin=synth.plan_inputs(d['root'],d['names'])
st=d['states'].numpy()
plan=engine.Plan(5,pin['parent'],pin['blen'],pin['leaf'],d['locus_offsets'],d['pi'],d['exch'],pin['T'],[10],[[5,15]],correction=pin['correction'])
for uval in (-1.0,0.0,1.0,2.0):
    u=np.full(3330,uval)
    f,g,h=plan.eval_columns(st,u)
    wf=wg=wh=0
    for l in range(10):
        sl=slice(l*333,(l+1)*333)
        for c in range(0,333,7):
            fo,go,ho=orc.column_curve(st[:,sl],pin['parent'],pin['blen'],pin['leaf'],d['pi'][l],d['exch'][l],c,np.array([uval]))
            wf=max(wf,abs(f[sl][c]-fo[0])); wg=max(wg,abs(g[sl][c]-go[0])); wh=max(wh,abs(h[sl][c]-ho[0]))
    print(uval,'max diff f,g,h',wf,wg,wh)
l=9;c=111;sl=slice(l*333,(l+1)*333)
for uval in (0.0,):
    f,g,h=plan.eval_columns(st,np.full(3330,uval))
    print('gpu',f[sl][c],g[sl][c],h[sl][c],'orc',orc.column_curve(st[:,sl],pin['parent'],pin['blen'],pin['leaf'],d['pi'][l],d['exch'][l],c,np.array([uval])))
