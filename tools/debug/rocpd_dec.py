import sqlite3, sys
c=sqlite3.connect(sys.argv[1]); cur=c.cursor()
rows=list(cur.execute("select name, start, end, scratch_size, lds_size from kernels order by start"))
for n,s,e,sc,l in rows[-40:]:
    if 'dec_ch' in n or 'backtrace' in n: print(f"dur {(e-s)/1e3:8.1f} scratch {sc} lds {l} {n[:70]}")
