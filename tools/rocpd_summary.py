"""Summarise a rocprofv3 rocpd sqlite database: per-kernel time stats and averaged PMC counters."""
import sqlite3, sys, collections

def tables(c):
    return {r[0].rsplit('_00', 1)[0] if False else r[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}

def find(c, prefix):
    for (n,) in c.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    raise KeyError(prefix)

def main(path, filt=""):
    c = sqlite3.connect(path)
    kd, ks = find(c, "rocpd_kernel_dispatch"), find(c, "rocpd_info_kernel_symbol")
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    rows = c.execute(f"select s.kernel_name, d.start, d.end, d.id from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = collections.defaultdict(list)
    for name, st, en, _ in rows:
        agg[name].append((en - st) / 1e3)
    print("kernel,calls,avg_us,min_us,max_us,total_us")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name[:90]
        print(f"{short},{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v):.1f}")
    try:
        pe, ip = find(c, "rocpd_pmc_event"), find(c, "rocpd_info_pmc")
        q = f"""select s.kernel_name, p.name, avg(e.value), count(*) from {pe} e join {ip} p on e.pmc_id = p.id
                join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name"""
        res = c.execute(q).fetchall()
        if res:
            print("\nkernel,counter,avg_value,samples")
            for name, cn, v, n in res:
                if filt in name:
                    print(f"{name[:60]},{cn},{v:.1f},{n}")
    except Exception as e:
        print("no pmc:", e)

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
