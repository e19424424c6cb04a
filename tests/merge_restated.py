"""Codecs::Google::IndexSession::merge (google_codec.cpp:186-438) restated over plain postings — the checker of tri_merge_google (tests only).
Per output term: the participants' documents in ascending order; a document that several participants hold comes from the MOST RECENT one (the
lowest participant index: `toAdvance[0]`, "first is always the most recent", :399); it is dropped when THAT participant's masked registry holds it
(:398) — every other participant's copy is skipped either way (:406-431); the kept document's hits and payloads are replayed into the encoder
(:324-366).  Pinned against the genuine reference by tests/golden/ref_merge.json (tests/test_golden_merge.py)."""
import numpy as np


def merge_term(lists, masked=None):
    """lists: per participant (most recent first) None or a list of (doc, hits) with hits = [(pos, payload_len, payload_value), ...];
    masked: per participant a set of documentIDs (or None).  Returns the merged [(doc, hits), ...]."""
    best = {}
    for p, lst in enumerate(lists):
        for doc, hits in lst or []:
            best.setdefault(int(doc), (p, hits))
    out = []
    for doc in sorted(best):
        p, hits = best[doc]
        if masked is not None and masked[p] is not None and doc in masked[p]:
            continue
        out.append((doc, hits))
    return out


def encoder_arrays(terms):
    """terms: per output term the merged [(doc, hits), ...] -> the arrays engine.host_encode_google / Device.encode_google take."""
    docs, freqs, pos, plen, pval, tf = [], [], [], [], [], [0]
    for lst in terms:
        for doc, hits in lst:
            docs.append(doc)
            freqs.append(len(hits))
            for h in hits:
                pos.append(int(h[0]))
                plen.append(int(h[1]))
                pval.append(int(h[2]))
        tf.append(len(docs))
    return (np.array(docs, np.uint32), np.array(freqs, np.uint32), np.array(pos, np.uint16), np.array(tf, np.uint64), np.array(plen, np.uint8), np.array(pval, np.uint64))
