import numpy as np, sys
sys.path.insert(0,'.')
from tapir_amd import synth, engine
from oracle import oracle as orc
d=synth.simulate(10,333,5,13)
pin=synth.plan_inputs(d['root'],d['names'])
st=d['states'].numpy()
plan=engine.Plan(5,pin['parent'],pin['blen'],pin['leaf'],d['locus_offsets'],d['pi'],d['exch'],pin['T'],[10],[[5,15]],correction=pin['correction'])
got=plan.site_rates(st)
for l in range(10):
    sl=slice(l*333,(l+1)*333)
    r=orc.site_rates(st[:,sl],pin['parent'],pin['blen'],pin['leaf'],d['pi'][l],d['exch'][l])
    bad=np.flatnonzero(got['flag'][sl]!=r['flag'])
    for c in bad:
        print(l,c,'gpu flag',got['flag'][sl][c],'rate',got['rate'][sl][c],'lnl',got['lnl'][sl][c],'| orc flag',r['flag'][c],r['rate'][c],r['lnl'][c], st[:,l*333+c])
print('evals',plan.last_eval_count())
