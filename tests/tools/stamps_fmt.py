import sys
prev=None; out=[]
for ln in sys.stdin:
    p=ln.split()
    if len(p)!=2: continue
    try: v=float(p[1])
    except: continue
    if p[0].startswith('sy.') or p[0] in ('rm.sal','cv.out'):
        out.append("%s=%.0f"%(p[0],v))
print(" ".join(out))
