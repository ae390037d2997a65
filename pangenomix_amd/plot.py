"""Mean pan/core curves: mirror of the reference's plot.calculate_mean (plot.py:5-43).
The column means are the data; the figure is drawn only when matplotlib is installed."""


def calculate_mean(df_pan_core, jpgName=None):
    """One-row DataFrame of the column means (Pan1..PanS, Core1..CoreS); with `jpgName` and matplotlib
    available, `<jpgName>_plot.png` as the reference draws it."""
    import pandas as pd
    mean_values = df_pan_core.mean()
    mean_df = pd.DataFrame([mean_values], columns=df_pan_core.columns)
    if jpgName is not None:
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except ImportError:
            print('matplotlib is not installed: no figure written')
            return mean_df
        half = mean_df.shape[1] // 2
        pan = mean_df.iloc[:, :half].T.reset_index(drop=True)
        core = mean_df.iloc[:, half:].T.reset_index(drop=True)
        pan.index, core.index = pan.index + 1, core.index + 1
        plt.plot(pan, label='Pangenome size')
        plt.plot(core, label='Core gene size')
        plt.xlabel('number of genomes')
        plt.ylabel('number of genes')
        plt.legend()
        plt.savefig('%s_plot.png' % jpgName)
    return mean_df
