"""MI355X-native neural-BSDF importance sampler (the sample()/pdf() hot path of
fzy28/BSDF_diffusion_sampling), see DESIGN.md.

Modules (everything below the tensors runs in libbsdfd.so, include/bsdfd.h; no CPU fallback):
  mlp_brdf_sampling   the reference's four operator functions (network_sampling_* / network_pdf_*)
  brdf_measured_disk, brdf_measured_spherical, bsdf_myresult
                      the reference's three ``MyBSDF`` plugin classes (sample / eval / pdf / eval_pdf)
  sampler             FlowSampler: the ctypes host of the fused flow kernels
  model, weights      reference-named weight containers, the neutral .bsdfw weight format
  materials           MaterialTable: material-tagged wavefronts (bucketing + segmented launches)
  measured            MeasuredBSDF: the ground-truth evaluator behind eval() (RGL tensor files)
  encoding            positional_encoding_1 as a stand-alone pass
  wavefront           WavefrontRenderer / ArrayRenderer: the render loop around the plugin calls
  sharding            query / image-row sharding over one process per GPU
  render_cli          ``python -m bsdf_diffusion_sampling_amd.brdf_measured_disk ...``
  mitsuba_adapter     optional ``mi.BSDF`` subclasses (needs Mitsuba; untested here)
"""
from . import weights  # noqa: F401

__all__ = ["weights"]
__version__ = "0.1.0"
