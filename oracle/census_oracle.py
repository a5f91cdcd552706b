"""CPU restatement of the reference's VQ-index analysis bookkeeping.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under
kindergarten-vq-vae_amd/); never the thing measured or shipped.

Follows /root/reference/analyses/unsupervised_vq_disentanglement/unsupervised_vq_disentanglement.py line by line:
  :166-200  the sentence -> word -> token walk (`census_walk`)
  :208-235  the three results (`census_results`)
PARITY UNPINNED: the reference script is a top-level program that needs the dSentences corpus, a trained checkpoint and a
tokenizer fetched by name (none exist offline), it has no tests and no fixtures, and it cannot be imported without running all
of that.  The walk below is a restatement from reading; the tests pin the HIP kernel and the host span index to IT, and it to
hand-worked cases (the commented example of the reference, :133-139, gives the intended word -> tokens -> codes pairing).
Pure Python loops on purpose (small cases only)."""


def census_walk(sentences, indices, word_token_count, n_codes, words_of_interest):
    """sentences: list[str]; indices: per sentence the flat list of code ids (v_i.flatten().tolist(), :168);
    word_token_count(word) -> number of tokens of the word tokenised alone (:174)."""
    words_of_interest_vq_distrib = {k: [] for k in words_of_interest}        # :112-114
    vq_words_distrib = {k: [] for k in range(n_codes)}                        # :115-117 (range(9) = VQ_N_E there)
    seen_v_is = set()                                                          # :141
    for s, v_i in zip(sentences, indices):                                     # :166
        s_i = 0                                                                # :171
        for word in s.split(" "):                                              # :173
            n_tokens = word_token_count(word)                                  # :174
            v_is = []
            for j in range(n_tokens):                                          # :178
                v_is.append(v_i[s_i + j])
                vq_words_distrib[v_is[-1]].append(word)                        # :181 "done on all words"
                seen_v_is.add(v_i[s_i + j])
            s_i += n_tokens                                                    # :185
            if word in words_of_interest and v_is:                             # :189-197
                words_of_interest_vq_distrib[word].append(v_is[0])
    return words_of_interest_vq_distrib, vq_words_distrib, seen_v_is


def census_results(words_of_interest_vq_distrib, vq_words_distrib, seen_v_is, n_codes):
    histograms = {}
    for word in words_of_interest_vq_distrib:                                  # :210-220
        histogram = {k: 0 for k in range(n_codes)}
        for ind in set(words_of_interest_vq_distrib[word]):
            histogram[ind] = words_of_interest_vq_distrib[word].count(ind)
        histograms[word] = histogram
    words_of_code = {k: sorted(set(v)) for k, v in vq_words_distrib.items()}   # :225-227 (list(set(v)): order left open there)
    return {"populated": set(seen_v_is), "histograms": histograms, "words_of_code": words_of_code}
