def master_only(fn):          # single process: every rank is the master
    return fn
