#!/usr/bin/env python3
"""
Builds tests/golden/xlmr_style_unigram_tokenizer.json + xlmr_style_unigram_expected.json: a SMALL SentencePiece-unigram
tokenizer assembled the way XLM-RoBERTa's (BGE-M3's) tokenizer.json is -- the file HIP_TOKENIZER_FILE points at in a
deployment (the real one cannot be shipped offline):

  normalizer      Sequence[ NFKC, Replace(" {2,}" -> " ") ]      (XLM-R: precompiled NFKC-based charsmap + the same Replace)
  pre_tokenizer   Sequence[ WhitespaceSplit, Metaspace("▁", prepend) ]
  model           Unigram, vocabulary = <s> <pad> </s> <unk> (ids 0..3: the fairseq layout, i.e. SentencePiece ids + 1 with
                  <s>/<pad>/</s> in front) followed by the learned pieces; unk_id = 3; <mask> appended last
  post_processor  TemplateProcessing  "<s> $A </s>"  /  "<s> $A </s> </s> $B </s>"
  decoder         Metaspace

Expected ids are produced by the `tokenizers` library itself (the implementation sentence-transformers runs under the
reference's HuggingFaceEmbeddings, rag/providers/hf/embeddings.py:32-35) in the build container; the test
(tests/test_overlay_cpu.py) pins the product's FileTokenizer against them.  Deterministic: fixed corpus, fixed trainer
settings.  Run from the repo root:  python tests/golden/make_unigram_fixture.py
"""
import json
import os

from tokenizers import Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, processors, trainers

HERE = os.path.dirname(os.path.abspath(__file__))

CORPUS = [
    "Invoice number 2024-0117 issued to the customer on 12 March 2024.",
    "Total amount due: 1,250.00 EUR including 19% VAT, payable within 30 days.",
    "Bank transfer details: IBAN DE89 3704 0044 0532 0130 00, reference INV-0117.",
    "The payment terms and the delivery address are listed on page two of the contract.",
    "Rechnung Nr. 0117: Gesamtbetrag 1.250,00 EUR, zahlbar innerhalb von 30 Tagen.",
    "Die Lieferadresse und die Zahlungsbedingungen stehen auf Seite zwei des Vertrags.",
    "Facture n° 0117 : montant total 1 250,00 EUR, TVA incluse, à régler sous 30 jours.",
    "L'adresse de livraison et les conditions de paiement figurent à la page deux.",
    "Hóa đơn số 0117: tổng số tiền 1.250,00 EUR, thanh toán trong vòng 30 ngày.",
    "Địa chỉ giao hàng và điều khoản thanh toán được ghi ở trang hai của hợp đồng.",
    "請求書番号0117：合計金額は1,250.00ユーロ、30日以内にお支払いください。",
    "Счёт № 0117: общая сумма 1 250,00 евро, оплата в течение 30 дней.",
    "retrieval augmented generation answers questions from the indexed pages of a document",
    "the embedding model maps a query and every chunk of text to a vector of 1024 numbers",
] * 3

TEXTS = [
    "Invoice total amount due",
    "  the   payment\tterms\n of the contract  ",              # runs of whitespace, tab, newline, leading / trailing
    "ＩＮＶＯＩＣＥ №０１１７ ﬁnal",                               # full-width letters, numero sign, fi ligature: NFKC folds them
    "Zahlungsbedingungen für die Lieferadresse",
    "tổng số tiền thanh toán",
    "請求書番号0117",
    "xyzzy qwertyuiop 🙂",                                       # out-of-vocabulary pieces -> <unk> (id 3)
    "",
    "a",
]
PAIRS = [("payment terms", "The payment terms and the delivery address are listed on page two."),
         ("tổng số tiền", "Hóa đơn số 0117: tổng số tiền 1.250,00 EUR"),
         ("", "bank transfer")]


def build() -> Tokenizer:
    tk = Tokenizer(models.Unigram())
    tk.normalizer = normalizers.Sequence([normalizers.NFKC(), normalizers.Replace(Regex(" {2,}"), " ")])
    tk.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(),
                                                pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")])
    trainer = trainers.UnigramTrainer(vocab_size=400, special_tokens=["<s>", "<pad>", "</s>", "<unk>"], unk_token="<unk>",
                                      show_progress=False)
    tk.train_from_iterator(CORPUS, trainer)
    tk.add_special_tokens(["<mask>"])
    tk.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                      special_tokens=[("<s>", 0), ("</s>", 2)])
    tk.decoder = decoders.Metaspace(replacement="▁", prepend_scheme="always")
    return tk


def main():
    tk = build()
    path = os.path.join(HERE, "xlmr_style_unigram_tokenizer.json")
    tk.save(path)
    tk = Tokenizer.from_file(path)
    vocab = tk.get_vocab()
    assert [vocab[t] for t in ("<s>", "<pad>", "</s>", "<unk>")] == [0, 1, 2, 3]
    exp = {"note": "ids produced by the `tokenizers` library in the build container from xlmr_style_unigram_tokenizer.json",
           "vocab_size": tk.get_vocab_size(), "mask_id": vocab["<mask>"],
           "single": [{"text": t, "ids_no_special": tk.encode(t, add_special_tokens=False).ids,
                       "ids_template": tk.encode(t, add_special_tokens=True).ids,
                       "tokens": tk.encode(t, add_special_tokens=False).tokens} for t in TEXTS],
           "pairs": [{"a": a, "b": b, "ids_template": tk.encode(a, b, add_special_tokens=True).ids} for a, b in PAIRS]}
    with open(os.path.join(HERE, "xlmr_style_unigram_expected.json"), "w", encoding="utf-8") as f:
        json.dump(exp, f, ensure_ascii=False, indent=1)
    print("vocab", exp["vocab_size"], "examples", len(TEXTS), len(PAIRS))
    for e in exp["single"][:4]:
        print(repr(e["text"]), e["tokens"])


if __name__ == "__main__":
    main()
