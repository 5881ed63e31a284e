"""Vocabulary of the text channels (reference utils/vocab_new.py:8-70; the dataset's word2id,
utils/Multi_GCN_Co_att_dataset.py:95-100): words of the train split with frequency >= min_count in order of
first appearance, 'PAD' at index 0 and 'UNK' at index 1; unknown words map to 'UNK'."""
import os


def get_vocab(vocab_root_path, text_min_count):
    """Read <root>/vocab/vocab-<N>.txt (one word per line, as the reference writes it)."""
    with open(os.path.join(vocab_root_path, 'vocab', 'vocab-%s.txt' % text_min_count)) as f:
        return f.read().split('\n')


def build_vocab(train_all_text, text_min_count, vocab_root_path=None):
    """Same list as vocab_new.build_vocab, in O(total words) (the reference scans a list per word)."""
    freq = {}
    for text in train_all_text:
        for word in text.split(' '):
            freq[word] = freq.get(word, 0) + 1          # dict keeps first-appearance order
    results = ['PAD', 'UNK'] + [w for w, c in freq.items() if c >= text_min_count]
    if vocab_root_path is not None:
        os.makedirs(os.path.join(vocab_root_path, 'vocab'), exist_ok=True)
        with open(os.path.join(vocab_root_path, 'vocab', 'vocab-%s.txt' % text_min_count), 'w') as f:
            f.write('\n'.join(results))
    return results


def get_vocab_list(data_root_path, vocab_root_path, text_min_count, train_all_text=None):
    try:
        return get_vocab(vocab_root_path, text_min_count)
    except FileNotFoundError:
        if train_all_text is None:
            import json
            train_all_text = []
            with open(os.path.join(data_root_path, 'all_anno_json', 'train_all_anno.json')) as f:
                for line in f:
                    train_all_text.append(json.loads(line)['text'])
        return build_vocab(train_all_text, text_min_count, vocab_root_path)


class Word2Id:
    def __init__(self, vocab):
        self.d = {w: i for i, w in enumerate(vocab)}
        self.unk = self.d['UNK']
        self.pad = self.d['PAD']

    def __call__(self, word):
        return self.d.get(word, self.unk)
