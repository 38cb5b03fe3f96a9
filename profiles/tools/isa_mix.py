import sys
S=sys.argv[1] if len(sys.argv)>1 else '/tmp/hnrf_mlp_f16-hip-amdgcn-amd-amdhsa-gfx950.s'
lines=open(S).read().split('\n')
labels=[(i,l.split(':')[0]) for i,l in enumerate(lines) if l.startswith('_ZN4hnrf') and ':' in l]
for k,(i,name) in enumerate(labels):
    j=next(x for x in range(i,len(lines)) if lines[x].startswith('.Lfunc_end'))
    body=lines[i:j]
    c={'v':0,'m':0,'d':0,'g':0,'s':0,'w':0,'n':0}
    for l in body:
        t=l.split()
        if not t: continue
        o=t[0]
        if o.startswith('v_mfma'): c['m']+=1
        elif o.startswith('v_'): c['v']+=1
        elif o.startswith('ds_'): c['d']+=1
        elif o.startswith('global_') or o.startswith('buffer_'): c['g']+=1
        elif o.startswith('s_'):
            c['s']+=1
            if o=='s_waitcnt': c['w']+=1
            if o=='s_nop': c['n']+=1
    if c['m']>100:
        scr=[l for l in lines[j:j+60] if 'ScratchSize' in l]
        print('%-46s VALU %5d DS %4d VMEM %4d SALU %5d (wait %4d nop %4d) MFMA %4d -> %.2f per MFMA %s' % (name[9:55],c['v'],c['d'],c['g'],c['s'],c['w'],c['n'],c['m'],(c['v']+c['d']+c['g']+c['s'])/c['m'], scr[0].strip() if scr else ''))
