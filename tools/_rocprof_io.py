"""Read rocprofv3 output either as CSV (`--output-format csv`) or as the rocpd SQLite file ROCm 7.2 writes by default."""

from __future__ import annotations

import csv
import glob
import os
import sqlite3


def _find(out_dir, pattern):
    return sorted(glob.glob(os.path.join(out_dir, "**", pattern), recursive=True))


def kernel_rows(out_dir):
    """[{Kernel_Name, Dispatch_Id, Start_Timestamp, End_Timestamp}] sorted by start time."""
    rows = []
    files = _find(out_dir, "*kernel_trace.csv")
    if files:
        for path in files:
            with open(path, newline="") as f:
                rows += [{"Kernel_Name": r["Kernel_Name"], "Dispatch_Id": int(r["Dispatch_Id"]),
                          "Start_Timestamp": int(r["Start_Timestamp"]), "End_Timestamp": int(r["End_Timestamp"])} for r in csv.DictReader(f)]
    else:
        dbs = _find(out_dir, "*_results.db")
        if not dbs:
            raise SystemExit(f"no kernel trace (csv or rocpd db) under {out_dir}")
        for path in dbs:
            con = sqlite3.connect(path)
            for name, disp, start, end in con.execute('select name, dispatch_id, start, "end" from kernels'):
                rows.append({"Kernel_Name": name, "Dispatch_Id": int(disp), "Start_Timestamp": int(start), "End_Timestamp": int(end)})
            con.close()
    rows.sort(key=lambda r: r["Start_Timestamp"])
    return rows


def counter_rows(out_dir):
    """[{Kernel_Name, Dispatch_Id, Counter_Name, Counter_Value}] sorted by dispatch id."""
    rows = []
    files = _find(out_dir, "*counter_collection.csv")
    if files:
        for path in files:
            with open(path, newline="") as f:
                rows += [{"Kernel_Name": r["Kernel_Name"], "Dispatch_Id": int(r["Dispatch_Id"]), "Counter_Name": r["Counter_Name"],
                          "Counter_Value": float(r["Counter_Value"]), "Grid_Size": r.get("Grid_Size", ""),
                          "Duration_Ns": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if r.get("End_Timestamp") and r.get("Start_Timestamp") else 0}
                         for r in csv.DictReader(f)]
    else:
        dbs = _find(out_dir, "*_results.db")
        if not dbs:
            raise SystemExit(f"no counter collection (csv or rocpd db) under {out_dir}")
        for path in dbs:
            con = sqlite3.connect(path)
            # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch
            q = "select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name"
            for name, disp, counter, value in con.execute(q):
                rows.append({"Kernel_Name": name, "Dispatch_Id": int(disp), "Counter_Name": counter, "Counter_Value": float(value)})
            con.close()
    rows.sort(key=lambda r: r["Dispatch_Id"])
    return rows
