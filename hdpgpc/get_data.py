from hdpgpc_amd.get_data import *  # noqa: F401,F403
