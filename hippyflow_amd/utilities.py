"""MultiVector <-> dense numpy converters (hippyflow/utilities/mv_utilities.py:18-54): the layout
contract at the host/device boundary.  A dense array is (N, nvec), C-ordered, as saved in
``*_decoder.npy`` / ``POD_projector.npy``."""
import numpy as np

from .multivector import MultiVector


def mv_to_dense_local(multivector):
    return multivector.to_dense()


def mv_to_dense(multivector):
    """One process per GPU and no mesh partition on the device: gather_on_zero is the identity."""
    return multivector.to_dense()


def dense_to_mv_local(dense_array, dl_vector=None):
    """(N, nvec) array -> MultiVector; ``dl_vector`` (a template vector) is accepted for signature
    compatibility and only supplies the context."""
    return MultiVector.from_dense(np.asarray(dense_array, dtype=np.float64), ctx=getattr(dl_vector, "ctx", None))
