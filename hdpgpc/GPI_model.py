from hdpgpc_amd.GPI_model import *  # noqa: F401,F403
from hdpgpc_amd.GPI_model import matrix_normal_inv_wishart  # noqa: F401
