"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the OpenObj object-NeRF hot path (see oracle/objnerf_oracle.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (openobj_amd/) never does.
"""
