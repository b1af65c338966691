#!/usr/bin/env python3
"""Prints a rocprofv3 --stats kernel summary compactly: python tools/kstats.py <output dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
