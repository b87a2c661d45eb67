"""DotPredictor (mirror of the reference's model/predictors/dot_predictor.py:6-10)."""
from legommenders_amd import functional as F_hip
from legommenders_amd.model.predictors.base_predictor import BasePredictor


class DotPredictor(BasePredictor):
    def predict(self, user_embeddings, item_embeddings):
        return F_hip.rowdot(user_embeddings, item_embeddings)      # sum(u * i, dim=-1)
