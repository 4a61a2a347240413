"""Figures of the two drivers (matplotlib, headless "Agg" backend: the figure is written to a file).

Same content as the reference's figures -- run_iLQR_open_loop.py:115-145 (one panel per state and one for the
control, solid blue 'iLQR' curves, titles 'Optimal State Trajectories' / 'Optimal Control Input') and
run_iLQR_MPC.py:150-186 (closed-loop states with the dashed red target, black control) -- for any n_x, n_u."""
import numpy as np

PENDULUM_LABELS = ("Theta (rad)", "Theta_dot (rad/s)")
DOUBLE_PENDULUM_LABELS = ("Theta 1 (rad)", "Theta 2 (rad)", "Theta_dot 1 (rad/s)", "Theta_dot 2 (rad/s)")


def state_labels(n_x):
    return {2: PENDULUM_LABELS, 4: DOUBLE_PENDULUM_LABELS}.get(n_x, tuple(f"x[{i}]" for i in range(n_x)))


def _figure(n_rows):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig = plt.figure(figsize=(10, max(8.0, 2.7 * n_rows)), facecolor="w")
    return plt, fig


def open_loop_figure(path, tspan, X_bar, U_bar):
    """run_iLQR_open_loop.py:115-145.  X_bar (n_x, N+1), U_bar (n_u, N), tspan (N+1)."""
    X_bar, U_bar = np.asarray(X_bar), np.asarray(U_bar)
    n_x, n_u = X_bar.shape[0], U_bar.shape[0]
    plt, fig = _figure(n_x + 1)
    for i, lab in enumerate(state_labels(n_x)):
        ax = plt.subplot(n_x + 1, 1, i + 1)
        ax.plot(tspan, X_bar[i, :], "b-", linewidth=2, label="iLQR")
        if i == 0:
            ax.set_title("Optimal State Trajectories")
            ax.legend()
        ax.set_xlabel("Time (s)")
        ax.set_ylabel(lab)
        ax.grid(True)
    ax = plt.subplot(n_x + 1, 1, n_x + 1)
    for j in range(n_u):
        ax.plot(tspan[:-1], U_bar[j, :], "b-", linewidth=2, label="iLQR" if n_u == 1 else f"iLQR u[{j}]")
    ax.set_title("Optimal Control Input")
    ax.set_xlabel("Time (s)")
    ax.set_ylabel("Control (torque)")
    ax.grid(True)
    plt.tight_layout()
    fig.savefig(path)
    plt.close(fig)
    return path


def closed_loop_figure(path, tspan_sim, X_sim, U_sim, x_target):
    """run_iLQR_MPC.py:150-186.  X_sim (n_x, N_sim+1), U_sim (n_u, N_sim), tspan_sim (N_sim+1)."""
    X_sim, U_sim = np.asarray(X_sim), np.asarray(U_sim)
    n_x, n_u = X_sim.shape[0], U_sim.shape[0]
    plt, fig = _figure(n_x + 1)
    for i, lab in enumerate(state_labels(n_x)):
        ax = plt.subplot(n_x + 1, 1, i + 1)
        ax.plot(tspan_sim, X_sim[i, :], "b-", linewidth=2, label="Closed Loop")
        ax.axhline(x_target[i], color="r", linestyle="--", linewidth=2, label="Target")
        if i == 0:
            ax.set_title("Closed-Loop State Trajectories")
            ax.legend()
        ax.set_xlabel("Time (s)")
        ax.set_ylabel(lab)
        ax.grid(True)
    ax = plt.subplot(n_x + 1, 1, n_x + 1)
    for j in range(n_u):
        ax.plot(tspan_sim[:-1], U_sim[j, :], "k-", linewidth=2, label="Control Input")
    ax.set_title("Optimal Control Input")
    ax.set_xlabel("Time (s)")
    ax.set_ylabel("Control (torque)")
    ax.grid(True)
    plt.tight_layout()
    fig.savefig(path)
    plt.close(fig)
    return path
