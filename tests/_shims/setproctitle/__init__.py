def setproctitle(title):          # basicts/runners/base_runner.py:47
    return None
