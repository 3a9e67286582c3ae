"""Prompt templates used on the streaming path (mirror of the CHATML entries of the reference's
`longva/conversation.py:414-423,525-562`; only the Qwen templates the entry point selects:
`qwen_1_5` (answers, --conv-mode), `qwen_1_5_ego` (chunk captions), `qwen_1_5_caption`,
`qwen_1_5_summarize` (merge summaries)).  Other separator styles belong to other model families
and are out of scope."""
import dataclasses
from typing import List, Optional, Tuple


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Tuple[str, str] = ("<|im_start|>user", "<|im_start|>assistant")
    messages: List[List[Optional[str]]] = dataclasses.field(default_factory=list)
    sep: str = "<|im_end|>"
    version: str = "qwen"

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self) -> str:
        """CHATML rendering (reference conversation.py:85-95): system + sep, then role\\nmessage + sep per turn;
        an empty (None) message leaves the role header open for generation."""
        ret = "" if self.system == "" else self.system + self.sep + "\n"
        for role, message in self.messages:
            if message:
                if isinstance(message, tuple):
                    message, images = message
                    message = "<image>" * len(images) + message
                ret += role + "\n" + message + self.sep + "\n"
            else:
                ret += role + "\n"
        return ret

    def copy(self):
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            sep=self.sep, version=self.version)


_SYS = "<|im_start|>system\n"
conv_templates = {
    "qwen_1_5": Conversation(system=_SYS + "You are a helpful assistant."),
    "qwen_1_5_ego": Conversation(
        system=_SYS + "    You are a useful assistant. What you see is video from my first-person perspective "
                      "and you need to conduct multiple rounds of dialogue with me."),
    "qwen_1_5_caption": Conversation(
        system=_SYS + "    You are a useful visual assistant. Please describe what you see in this video in as much detail "
                      "as possible from a first-person perspective, including the surrounding environment, what objects "
                      "are there, etc.\n    PLEASE DO NOT GENERATE TEXT YOU ARE NOT SURE ABOUT."),
    "qwen_1_5_summarize": Conversation(system=_SYS + "    You are a helpful assistant."),
}
