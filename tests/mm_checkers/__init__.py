"""CHECKERS (test infrastructure, not product): the Python implementations that the native host code behind the C ABI
replaced -- the post-steps of align_frames_in_geometry and postprocess_geom_pair (postproc.py, postproc_flat.py,
api_python.py) and the geometry builder (py_builder.py).  They pass the reference's own unit tests restated in
tests/test_postproc.py / tests/test_refbuild.py, and tests/test_native_frames.py, test_property_host.py compare the native
code with them bit for bit.  The product reaches them only through the MM_PY_POSTPROC / MM_PY_BUILDER switches of the test
suite (multimoda_rs_amd.api._checker), which fail loudly when this package is not on sys.path."""
