"""hdpgpc/hdpgpc/util_plots.py: the result table the drivers print (util_plots.py:269-299).  Figures are presentation and
out of scope: plot_models_plotly is a no-op that says so."""
import numpy as np


def print_results(sw_gp, labels, N_0, error=False, purity=False):
    """Per cluster: histogram of the annotation labels of its members and the majority label; then the number of members
    whose label differs from their cluster's majority ("classification error")."""
    models = sw_gp.gpmodels[0]
    main_model = ["None"] * len(models)
    for i, gp in enumerate(models):
        vals, counts = np.unique([labels[j + N_0] for j in gp.indexes], return_counts=True)
        hist = "[" + ",".join(f"{v}-{c}" for v, c in zip(vals, counts)) + "]"
        mm = ""
        if len(counts) > 0:
            main_model[i] = vals[np.argmax(counts)]
            mm = ": MainModel: " + str(main_model[i])
        print('Model', (i + 1), mm, ':', hist)
    err = np.zeros(len(models))
    for m, gp in enumerate(models):
        err[m] = sum(1 for i in gp.indexes if labels[i + N_0] != main_model[m])
        if purity:
            print('Model', (m + 1), ': Purity: ', 1 - err[m] / len(gp.indexes))
    tot = int(err.sum())
    print(f"Classification error: {tot} / {sw_gp.T} -- {(tot / sw_gp.T):.5f}")
    if purity:
        print(f"Classification purity: {sw_gp.T - tot}/{sw_gp.T} -- {(1 - err.sum() / sw_gp.T):.5f}")
        return main_model, tot, sw_gp.T - tot
    if error:
        return main_model, tot
    return main_model


def plot_models_plotly(*args, save=None, **kwargs):
    """util_plots.py:725-794 draws the clusters with plotly / matplotlib: presentation, not part of this build (SURVEY.md
    section 2, row 12).  Every reference driver ends with this call, so it returns quietly instead of raising."""
    print("plot_models_plotly: figures are not part of the MI355X build" + (f" (nothing written to {save})" if save else ""))
    return None
