"""Prompt fragments (reference memory_bank/prompt_utils.py:37-43)."""


def only_related_prompt_dict_ego():
    return {"en": """
    Based on the current user's question, the most relevant historical contextual conversation records are: "{related_memory_content}".
    """}
