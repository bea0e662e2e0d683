"""Checker code for the HIP rasterization path: test infrastructure, never the product.

`oracle.port` wraps the plain-C restatement of the reference's algorithm
(oracle/lt_oracle.c); `oracle.ref` wraps libraries built from the reference's own
sources by oracle/build_ref.sh (present only where /root/reference was available at
build time).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.
"""
