#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv (or pmc csv) to readable width: keeps every row, cuts
kernel names to 90 characters.  usage: trim_stats.py in.csv > out.csv"""
import csv
import sys

rd = csv.reader(open(sys.argv[1]))
wr = csv.writer(sys.stdout)
for row in rd:
    wr.writerow([c if len(c) <= 90 else c[:87] + "..." for c in row])
