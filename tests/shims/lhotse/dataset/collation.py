def collate_features(*a, **k):
    raise NotImplementedError
