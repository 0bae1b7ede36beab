#!/usr/bin/env python3
"""Runs a command as a child and prints its peak resident memory (ru_maxrss of the children).  usage: python tools/peak_rss.py cmd args..."""
import resource
import subprocess
import sys
rc = subprocess.call(sys.argv[1:])
print("peak RSS of children: %.1f GiB, exit %d" % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 2**20, rc))
