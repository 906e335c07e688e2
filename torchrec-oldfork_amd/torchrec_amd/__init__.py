"""Host-side mirror of the reference's torchrec interfaces on the hot path (KJT, input/output
dists, sharded embedding bag collection, DLRM, train pipeline) for MI355X.  The reference's own
torchrec python can also sit on top of this repo's `fbgemm_gpu` package directly
(INTEGRATION.md); this package exists because the reference does not travel to the GPU box and
to remove the host syncs / extra copies listed in SURVEY.md §8f.2."""
import fbgemm_gpu as _fbgemm_gpu  # noqa: E402,F401  (sets HSA_ENABLE_IPC_MODE_LEGACY=0 before HIP initialises; loads the HIP library)
