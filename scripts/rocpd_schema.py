import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
for (name, typ) in cur.execute("select name, type from sqlite_master where type in ('table','view') order by name").fetchall():
    try:
        cols = [r[1] for r in cur.execute("pragma table_info('%s')" % name)]
        n = cur.execute("select count(*) from '%s'" % name).fetchone()[0]
    except Exception as e:
        print(name, typ, "ERR", e); continue
    print("==", typ, name, n, cols)
    if n and any(k in name for k in ("region", "marker", "kernel", "sample", "string", "event", "thread")) :
        for row in cur.execute("select * from '%s' limit 3" % name).fetchall():
            print("   ", tuple(str(x)[:60] for x in row))
