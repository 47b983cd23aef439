"""Runs __graft_entry__.smoke() as the driver does at round end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.smoke()
print("smoke ok")
