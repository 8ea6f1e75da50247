"""ORACLE — CPU restatement of the reference's decode hot path.  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package.  See oracle/ref_ops.py for the pinning status."""
