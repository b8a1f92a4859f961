"""Drop-in for the reference's ``modules.load_state.load_state`` (modules/load_state.py:4-15):
copy every checkpoint tensor whose key AND shape match, keep the net's own value otherwise and
print the reference's warning line."""
import collections


def load_state(net, checkpoint):
    source_state = checkpoint['state_dict']
    target_state = net.state_dict()
    merged = collections.OrderedDict()
    for key, value in target_state.items():
        src = source_state.get(key) if hasattr(source_state, "get") else None
        if src is not None and tuple(src.shape) == tuple(value.shape):
            merged[key] = src
        else:
            merged[key] = value
            print('[WARNING] Not found pre-trained parameters for {}'.format(key))
    net.load_state_dict(merged)
