import numpy as np, sys
sys.path.insert(0,'.')
from tapir_amd import synth, engine
from oracle import oracle as orc
d=synth.simulate(10,333,5,13)
pin=synth.plan_inputs(d['root'],d['names'])
st=d['states'].numpy()
plan=engine.Plan(5,pin['parent'],pin['blen'],pin['leaf'],d['locus_offsets'],d['pi'],d['exch'],pin['T'],[10],[[5,15]],correction=pin['correction'])
got=plan.site_rates(st)
lam,U,Ui,kap=plan.models()
for l in range(10):
    sl=slice(l*333,(l+1)*333)
    r=orc.site_rates(st[:,sl],pin['parent'],pin['blen'],pin['leaf'],d['pi'][l],d['exch'][l])
    ok=(r['flag']==0)
    rel=np.abs(got['rate'][sl]-r['rate'])/np.maximum(r['rate'],1e-12)
    for c in np.flatnonzero(ok&(rel>1e-9)):
        ug=np.log(got['rate'][sl][c]/kap[l]); uo=np.log(r['rate'][c]/kap[l])
        f,g,h=orc.column_curve(st[:,sl],pin['parent'],pin['blen'],pin['leaf'],d['pi'][l],d['exch'][l],c,np.array([ug,uo]))
        print(l,c,'rel',rel[c],'u gpu/orc',ug,uo,'g',g,'h',h,'dlnl',got['lnl'][sl][c]-r['lnl'][c], st[:,l*333+c])
