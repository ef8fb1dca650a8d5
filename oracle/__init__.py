"""oracle/: TEST INFRASTRUCTURE.  CPU restatement of the reference's hot-path arithmetic (eesen_oracle.c,
net.py) and a binding to the real reference compiled in place (oracle/_ref, refbind.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package — as the
checker / the timed CPU baseline, never as the product path.  eesen_amd/ must not import it.
"""
