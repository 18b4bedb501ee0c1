"""The stub model the reference's `train_epoch` / `test` were driven with when
`tests/golden/train_loop_vectors.npz` was generated (same definition as `StubModel` in
`tests/golden/make_golden.py`): `.obs_window` + a fixed per-node map of the flattened window."""
import torch


class StubModel(torch.nn.Module):
    def __init__(self, obs, W0, b0, device="cpu"):
        super().__init__()
        self.obs_window = int(obs)
        self.W = torch.nn.Parameter(torch.as_tensor(W0).clone().to(device))
        self.b = torch.nn.Parameter(torch.as_tensor(b0).clone().to(device))

    def forward(self, X, attention_threshold=0.0, **kw):
        return 0.5 * torch.tanh(X @ self.W.t() + self.b)


CASES = {
    "ar1_plain": dict(ar=1),
    "ar1_lat": dict(ar=1, lat=True),
    "ar2_static_forcing": dict(ar=2, lat=True, chan=True, static=[2], forcing=[4]),
    "ar3_all": dict(ar=3, lat=True, chan=True, smask=True, static=[2], forcing=[0, 4]),
    "ar3_noresidual": dict(ar=3, lat=True, static=[2], forcing=[4], residual=False),
    "ar5_capped": dict(ar=5, lat=True),
}
EVAL_CASES = {
    "eval_plain": dict(),
    "eval_all": dict(lat=True, smask=True, chan=True, static=[2], forcing=[0, 4]),
    "eval_noresidual": dict(lat=True, static=[2], residual=False),
}
