from hdpgpc_amd.GPI_HDP import *  # noqa: F401,F403
