"""LegoConfig (mirror of the reference's model/lego_config.py:82-256): hyper-parameters, component classes,
instantiation of item operator / user operator / predictor, vocabulary registration."""
from __future__ import annotations

from typing import Any, Dict, Optional, Type

from legommenders_amd.loader.column_map import ColumnMap
from legommenders_amd.loader.embedding_hub import EmbeddingHub
from legommenders_amd.model.operators.base_operator import BaseOperator
from legommenders_amd.model.predictors.base_predictor import BasePredictor


def combine_config(config: Dict[str, Any], **kwargs) -> Dict[str, Any]:
    """utils/function.py:31-52: fill in defaults for keys the user yaml did not set."""
    for k, v in kwargs.items():
        if k not in config:
            config[k] = v
    return config


class LegoConfig:
    cm: ColumnMap
    eh: EmbeddingHub
    item_operator: Optional[BaseOperator]
    user_operator: BaseOperator
    predictor: BasePredictor

    def __init__(self, hidden_size: int, user_config: dict, *, neg_count: int = 4,
                 item_hidden_size: Optional[int] = None, item_config: Optional[dict] = None,
                 predictor_config: Optional[dict] = None, use_neg_sampling: bool = True,
                 use_item_content: bool = True, use_fast_eval: bool = True, item_page_size: int = 0,
                 cache_page_size: int = 512, **kwargs):
        self.hidden_size = hidden_size
        self.item_hidden_size = item_hidden_size or hidden_size
        self.use_item_content = use_item_content
        self.item_config = item_config
        self.user_config = user_config
        self.predictor_config = predictor_config or {}
        self.use_neg_sampling = use_neg_sampling
        self.neg_count = neg_count
        self.item_page_size = item_page_size
        self.cache_page_size = cache_page_size
        self.use_fast_eval = use_fast_eval
        if self.use_item_content:
            self.item_config = self.item_config or {}

    def set_component_classes(self, item_operator_class: Type[BaseOperator], user_operator_class: Type[BaseOperator],
                              predictor_class: Type[BasePredictor]):
        self.item_operator_class = item_operator_class
        self.user_operator_class = user_operator_class
        self.predictor_class = predictor_class

    def set_item_ut(self, item_ut, item_inputs: list):
        self.item_ut, self.item_inputs = item_ut, item_inputs

    def set_user_ut(self, user_ut, user_inputs: list):
        self.user_ut, self.user_inputs = user_ut, user_inputs

    def set_column_map(self, cm: ColumnMap):
        self.cm = cm

    def set_embedding_hub(self, eh: EmbeddingHub):
        self.eh = eh

    def build_components(self):
        self.item_operator = None
        if self.use_item_content:
            item_config = self.item_operator_class.config_class(
                **combine_config(config=self.item_config, hidden_size=self.hidden_size, input_dim=self.item_hidden_size))
            self.item_operator = self.item_operator_class(config=item_config, target_user=False, lego_config=self)
        user_input_dim = self.item_operator.output_dim if self.use_item_content else self.item_hidden_size
        user_config = self.user_operator_class.config_class(
            **combine_config(config=self.user_config, hidden_size=self.hidden_size, input_dim=user_input_dim))
        if self.user_operator_class.flatten_mode:
            user_config.inputer_config["item_ut"] = self.item_ut
            user_config.inputer_config["item_inputs"] = self.item_inputs
        self.user_operator = self.user_operator_class(config=user_config, target_user=True, lego_config=self)
        if self.use_neg_sampling and not self.predictor_class.allow_matching:
            raise ValueError(f"{self.predictor_class.__name__} does not support negative sampling")
        if not self.use_neg_sampling and not self.predictor_class.allow_ranking:
            raise ValueError(f"{self.predictor_class.__name__} only supports negative sampling")
        predictor_config = self.predictor_class.config_class(
            **combine_config(config=self.predictor_config, hidden_size=self.hidden_size))
        self.predictor = self.predictor_class(config=predictor_config, lego_config=self)

    def register_inputer_vocabs(self):
        if self.use_item_content:
            for vocab in self.item_operator.inputer.get_vocabs():
                self.eh.register_vocab(vocab)
        for vocab in self.user_operator.inputer.get_vocabs():
            self.eh.register_vocab(vocab)
