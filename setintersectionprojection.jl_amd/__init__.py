"""sipx -- MI355X-native PARSDMM projection engine (host-side mirror of the reference API).

The directory name carries a dot, so load it with ``__graft_entry__.load_package()``
(registers the package as ``sipx``).  Everything numerical happens in ``libsipx.so``
(hand-written HIP for gfx950); importing this package without the built library, or
calling it without a GPU, raises -- there is no CPU fallback.
"""
from .host import (  # noqa: F401
    PARSDMM, PARSDMM_options, PARSDMM_precompute_distribute, PARSDMM_precompute_distribute_Minkowski, Context, SipxError, TDOperator,
    Projector, compgrid, default_PARSDMM_options, get_TD_operator, lib, log_type_PARSDMM,
    set_definitions, set_properties, setup_constraints, cds_spmv, CDS_MVp, LIB_PATH, EXPORTED_SYMBOLS,
    set_default_device, resample_nn, prox_l2s, clear_context_cache,
)
from .multilevel import (  # noqa: F401
    PARSDMM_multi_level, setup_multi_level_PARSDMM, constraint2coarse, interpolate_y_l,
)
