"""On-disk formats consumed downstream of the projectors.

* what the projectors save: ``AS_{n}{input_decoder_name}.npy`` + ``AS_{n}_d_GN.npy``, ``AS_{n}{output_decoder_name}.npy``
  + ``AS_{n}_d_NG.npy`` (activeSubspaceProjector.py:475-480,587-603), ``KLE_decoder.npy`` + ``KLE_d.npy``
  (KLEProjector.py:190-192), ``POD_projector.npy`` + ``POD_d.npy`` (PODProjector.py:382-384); all (N, r) / (r,) fp64;
* what the training side loads: ``AS_input_projector.npy``, ``AS_d_GN.npy``, ``AS_output_projector.npy``,
  ``AS_d_NG.npy``, ``KLE_projector.npy``, ``KLE_d.npy``, ``POD_projector.npy``, ``POD_d.npy``
  (applications/confusion/confusion_utilities.py:115-172).  The two naming schemes differ in the reference; the
  loader here accepts either.
* truncation by eigenvalue tolerance or fixed rank, re-orthogonalisation and rescaling of the chosen pair
  (confusion_utilities.py:174-225); the QR runs on the device.
"""
import glob
import os

import numpy as np

from .multivector import MultiVector


def _first_existing(data_dir, patterns):
    for pat in patterns:
        hits = sorted(glob.glob(os.path.join(data_dir, pat)))
        if hits:
            return hits[0]
    raise FileNotFoundError("none of %s found in %s" % (patterns, data_dir))


def _truncate(P, d_path_patterns, data_dir, tolerance, fixed_rank):
    if fixed_rank > 0:
        return P[:, :fixed_rank]
    d = np.load(_first_existing(data_dir, d_path_patterns))
    return P[:, np.where(d > tolerance)[0]]


def get_projectors(data_dir, as_input_tolerance=1e-4, as_output_tolerance=1e-4, kle_tolerance=1e-4, pod_tolerance=1e-4,
                   fixed_input_rank=0, fixed_output_rank=0, mixed_output=True, verbose=False):
    """Dictionary {'AS_input', 'AS_output', 'KLE', 'POD'} of (N, r) arrays truncated by tolerance or fixed rank.
    Projectors that are absent from ``data_dir`` are skipped."""
    spec = {
        'AS_input': (['AS_input_projector.npy', 'AS_*_input_decoder.npy'], ['AS_d_GN.npy', 'AS_*_d_GN.npy'], as_input_tolerance, fixed_input_rank),
        'AS_output': (['AS_output_projector.npy', 'AS_*_output_decoder.npy'], ['AS_d_NG.npy', 'AS_*_d_NG.npy'], as_output_tolerance, fixed_output_rank),
        'KLE': (['KLE_projector.npy', 'KLE_decoder.npy'], ['KLE_d.npy'], kle_tolerance, fixed_input_rank),
        'POD': (['POD_projector.npy'], ['POD_d.npy'], pod_tolerance, fixed_output_rank),
    }
    out = {}
    for key, (p_pat, d_pat, tol, fixed) in spec.items():
        try:
            P = np.load(_first_existing(data_dir, p_pat))
        except FileNotFoundError:
            continue
        if verbose:
            print(key, 'projector shape before truncation = ', P.shape)
        out[key] = _truncate(P, d_pat, data_dir, tol, fixed)
        if verbose:
            print(key, 'projector shape after truncation = ', out[key].shape)
    return out


def _orthonormalize(P):
    mv = MultiVector.from_dense(np.ascontiguousarray(P))
    mv.orthogonalize()
    return mv.to_dense()


def modify_projectors(projectors, input_subspace, output_subspace, seed=0):
    """Orthogonalise and rescale the chosen input/output projector pair for the network's first/last layers.
    Input: Q / (N/(32 r) * ||Q||_F); output: Q / ||Q||_F; 'random' draws a Gaussian basis of the same shape."""
    assert input_subspace in ['kle', 'as', 'random']
    assert output_subspace in ['pod', 'as', 'random']
    rng = np.random.default_rng(seed)
    if input_subspace == 'random':
        input_projector = _orthonormalize(rng.standard_normal(projectors['KLE'].shape))
    else:
        input_projector = _orthonormalize(projectors['KLE'] if input_subspace == 'kle' else projectors['AS_input'])
    scale_in = float(input_projector.shape[0]) / (32 * float(input_projector.shape[-1]))
    input_projector = input_projector / (scale_in * np.linalg.norm(input_projector))
    if output_subspace == 'random':
        output_projector = rng.standard_normal(projectors['POD'].shape)
        output_projector /= np.linalg.norm(output_projector)
    else:
        output_projector = _orthonormalize(projectors['POD'] if output_subspace == 'pod' else projectors['AS_output'])
        output_projector = output_projector / np.linalg.norm(output_projector)
    return input_projector, output_projector


def spectrum_plot(lambdas, axis_label=('i', r'$\lambda$', 'Spectrum'), ylims=None, out_name=None):
    """The eigenvalue plot the reference's projectors leave next to their arrays (utilities/plotting.py:18-50: the values
    above 1e-10 on a logarithmic axis, saved to ``out_name``).  Host-side cosmetics: drawn with matplotlib when it is
    installed (headless ``Agg`` canvas), silently skipped -- ``None`` returned -- when it is not."""
    try:
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
    except Exception:          # noqa: BLE001 -- no matplotlib / no usable backend: the arrays are what matters
        return None
    lam = np.asarray(lambdas, dtype=np.float64)
    lam = lam[lam > 1e-10]
    fig, ax = plt.subplots(figsize=(10, 5))
    ax.semilogy(np.arange(lam.size), lam)
    ax.set_xlabel(axis_label[0], fontsize=30)
    ax.set_ylabel(axis_label[1], fontsize=35)
    ax.set_title(axis_label[2], fontsize=35)
    if ylims is not None:
        ax.set_ylim(ylims)
    ax.tick_params(axis='both', which='both', labelsize=25)
    for tick in ax.get_yticklabels():
        tick.set_rotation(90)
    if out_name is not None:
        fig.savefig(out_name, bbox_inches='tight')
    plt.close(fig)
    return fig


def singular_values_plot(s, s_std, title='Average singular values with std', outname='out_plot.pdf'):
    """Mean singular values of the per-sample Jacobians with a band of one standard deviation, the PDF the reference writes next
    to ``J_on_proc*.npz`` (utilities/plotting.py:135-160, called at activeSubspaceProjector.py:880-883, 898-901).  Optional
    cosmetics like ``spectrum_plot``: ``None`` without matplotlib."""
    try:
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
    except Exception:          # noqa: BLE001
        return None
    s, s_std = np.asarray(s, dtype=np.float64), np.asarray(s_std, dtype=np.float64)
    fig, ax = plt.subplots()
    idx = np.arange(1, s.size + 1)
    ax.semilogy(idx, s)
    ax.fill_between(idx, s - s_std, s + s_std, alpha=0.2)
    ax.set_xlabel('i', fontsize=20)
    ax.set_ylabel(r'$\sigma_i$', fontsize=20)
    ax.set_title(title, fontsize=20)
    ax.grid()
    fig.tight_layout()
    fig.savefig(outname)
    plt.close(fig)
    return fig
