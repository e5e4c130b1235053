from hdpgpc_amd.amtgp_warping_system import *  # noqa: F401,F403
