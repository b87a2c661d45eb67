"""Dataset column schema (mirror of the reference's loader/column_map.py:24-68)."""


class ColumnMap:
    def __init__(self, history_col="history", item_col="item_id", label_col="click", user_col="user_id",
                 group_col="user_id", neg_col=None):
        self.history_col = history_col
        self.item_col = item_col
        self.label_col = label_col
        self.group_col = group_col
        self.user_col = user_col
        self.neg_col = neg_col
        self.mask_col = "__clicks_mask__"
