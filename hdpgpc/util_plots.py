from hdpgpc_amd.util_plots import *  # noqa: F401,F403
