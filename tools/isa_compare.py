import re, sys
def funcs(path):
    out, cur, name = {}, None, None
    for ln in open(path):
        m = re.match(r'^(_Z[\w]+|\.L_Z[\w]+):', ln)
        if m:
            name = m.group(1); cur = []; out[name] = cur; continue
        if cur is None: continue
        if ln.startswith('.Lfunc_end'):
            cur = None; continue
        s = ln.split(';')[0].rstrip()
        s = re.sub(r'\.LBB\d+_', '.LBB_', s)
        s = re.sub(r'\.Ltmp\d+', '.Ltmp', s)
        if not s.strip() or s.strip().startswith('.'): 
            if not re.match(r'^\.LBB_', s.strip()): continue
        cur.append(s)
    return out
a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
rc = 0
for k in sorted(set(a) | set(b)):
    if k not in a: print('only in B:', k, len(b[k])); continue
    if k not in b: print('only in A:', k, len(a[k])); continue
    same = a[k] == b[k]
    print(('SAME ' if same else 'DIFF '), k, len(a[k]), len(b[k]))
    if not same:
        rc = 1
        import difflib
        d = list(difflib.unified_diff(a[k], b[k], lineterm='', n=0))
        print('   ', len(d), 'diff lines'); 
        for x in d[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]: print('   ', x)
sys.exit(rc)
