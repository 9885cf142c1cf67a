"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's render_rays hot path (see oracle/field.py and
oracle/hashgrid.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package, and only as the checker / reported CPU
baseline.  Nothing under morpheus_amd/ imports it; the product path fails loudly
when the HIP library is missing instead of falling back to this code.
"""
