from hdpgpc_amd.GPI import *  # noqa: F401,F403
