"""Import names of the reference (`import hdpgpc.GPI_HDP as hdpgp`, `from hdpgpc.get_data import compute_estimators_LDS`,
...) bound to the MI355X build: every module here re-exports its counterpart in hdpgpc_amd (Face 1)."""
