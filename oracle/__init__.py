"""oracle/ -- CPU restatements of the reference's hot-path arithmetic.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
path (dreamwaltz-g_amd/) never does.
"""
