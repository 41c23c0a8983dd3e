"""syconn_amd -- MI355X-native drop-in for SyConn's chunked dense 3D-CNN prediction path.

Public surface (same names as the reference):
  syconn_amd.handler.prediction: Predictor, dense_predicton_helper, dense_predictor, predict_dense_to_kd
  syconn_amd.exec.exec_dense_prediction: predict_myelin / _synapsetype / _cellorganelles / _er / _golgi
  syconn_amd.handler.basics: chunkify, kd_factory
  syconn_amd.global_params: wd, config
Compute: libsyconn_dense_hip.so (C ABI in include/syconn_dense.h), hand-written gfx950 kernels; no CPU fallback.
"""
__version__ = '0.1'
