"""Per-kernel PMC counter summary from rocprofv3 --pmc result databases.
Usage: python tools/pmc_summary.py gpurun_out/prof_r01_pmc_*/ > profiles/r01_pmc.txt"""
import glob
import os
import sqlite3
import sys


def main(dirs):
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            db = sqlite3.connect(path)
            cur = db.cursor()
            try:
                rows = cur.execute(
                    "select k.kernel_name, p.name, count(*), sum(e.value), avg(e.value) "
                    "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                    "join rocpd_kernel_dispatch kd on e.event_id = kd.event_id "
                    "join rocpd_info_kernel_symbol k on kd.kernel_id = k.id "
                    "group by k.kernel_name, p.name order by sum(e.value) desc").fetchall()
            except Exception as ex:  # schema differences: dump what is there
                print("# %s: %s" % (path, ex))
                continue
            print("# %s" % path)
            print("%-70s %-28s %8s %18s %18s" % ("kernel", "counter", "launches", "sum", "avg/launch"))
            for r in rows:
                if "lcp" not in r[0]:
                    continue
                print("%-70s %-28s %8d %18.1f %18.1f" % (r[0][:70], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main(sys.argv[1:])
